import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bcftools_amd import abi, synth, engine
from tests.helpers import indeldrv
n_smpl = 30
b = synth.indel_batch(72, 16, n_smpl, depth=12.0)
c0 = engine.Context(abi.default_cfg(n_smpl, max_sites=16, max_reads=64))
def load(path):
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[:8], np.int64)[0])
    o = 8
    s1 = np.frombuffer(raw[o:o+4*n], np.int32); o += 4*n
    s2 = np.frombuffer(raw[o:o+4*n], np.int32); o += 4*n
    k = np.frombuffer(raw[o:o+4*n], np.uint32); o += 4*n
    pj = np.frombuffer(raw[o:o+16*n], np.dtype([("ref_off","u4"),("q8","u4"),("l_ref","u2"),("l_query","u2"),("eff","u2"),("pad","u2")]))
    return s1, s2, k, pj
runs = []
for r in range(4):
    os.environ["BCFGPU_DUMP_SCORES"] = "/tmp/other.bin"
    indeldrv.gap_prep_gpu(c0, synth.indel_batch(70 + r, 16, n_smpl, depth=12.0))
    os.environ["BCFGPU_DUMP_SCORES"] = "/tmp/sc%d.bin" % r
    g = indeldrv.gap_prep_gpu(c0, b)[0]
    if r == 0: g0 = g
    print("aux diffs vs run 0:", int((g["aux"] != g0["aux"]).sum()))
    runs.append(load("/tmp/sc%d.bin" % r))
s1, s2, k, pj = runs[0]
print("jobs", len(s1), "classes", np.unique(k >> 13, return_counts=True))
for r in range(1, 4):
    t1, t2, kk, pp = runs[r]
    d1 = np.nonzero(t1 != s1)[0]; d2 = np.nonzero(t2 != s2)[0]
    print("run", r, "s1 diffs", len(d1), "s2 diffs", len(d2), "keys equal", np.array_equal(k, kk))
    for d in d2[:12]:
        print("  job", d, "cls", k[d] >> 13, "pj", pj[d], "s1", s1[d] >> 8, t1[d] >> 8, "s2", s2[d] >> 8, t2[d] >> 8)
    if len(d2):
        print("  classes of s2 diffs", np.unique(k[d2] >> 13, return_counts=True), "lq range", pj["l_query"][d2].min(), pj["l_query"][d2].max())
