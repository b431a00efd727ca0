"""How long host/bcfgpu_sam takes on a region of real-sized input, and where: N samples x 30x of 100-base reads over a region
of a random contig, written as SAM files, run with and without BAQ.  python tools/sam_driver_probe.py [--samples 40] [--kb 60]"""
import argparse, os, subprocess, sys, tempfile, time
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=40)
ap.add_argument("--kb", type=int, default=60)
ap.add_argument("--depth", type=float, default=30.0)
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(root, "host", "bcfgpu_sam")
rng = np.random.default_rng(1)
L = a.kb * 1000
ref = rng.integers(0, 4, L)
d = tempfile.mkdtemp()
fa = os.path.join(d, "ref.fa")
with open(fa, "w") as f:
    f.write(">chr1\n")
    s = "".join("ACGT"[i] for i in ref)
    for i in range(0, L, 60):
        f.write(s[i:i + 60] + "\n")
refs = np.array(list(s))
files = []
t0 = time.time()
nreads = 0
for k in range(a.samples):
    p = os.path.join(d, "s%03d.sam" % k)
    files.append(p)
    n = int(L * a.depth / 100)
    pos = np.sort(rng.integers(0, L - 100, n))
    with open(p, "w") as f:
        f.write("@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chr1\tLN:%d\n@RG\tID:s%d\tSM:s%d\n" % (L, k, k))
        for i, q in enumerate(pos):
            seq = refs[q:q + 100].copy()
            e = np.flatnonzero(rng.random(100) < 0.005)
            seq[e] = np.array(list("ACGT"))[rng.integers(0, 4, len(e))]
            qual = "".join(chr(33 + int(x)) for x in rng.choice([11, 25, 37, 40], 100))
            f.write("r%d\t%d\tchr1\t%d\t60\t100M\t*\t0\t0\t%s\t%s\tRG:Z:s%d\n" % (i, 16 * int(rng.integers(0, 2)), q + 1, "".join(seq), qual, k))
    nreads += n
print("wrote %d reads of %d samples over %d kb in %.0f s" % (nreads, a.samples, a.kb, time.time() - t0), flush=True)
for opts, om in ((["-B"], "u"), (["-B"], "v"), ([], "u"), (["-B", "--tile", "4096"], "u"), (["-B", "--tile", "65536"], "u")):
    t0 = time.time()
    p = subprocess.run([exe, "--timing"] + opts + ["-O", om, "-o", os.path.join(d, "out." + om), "-f", fa, "-r", "chr1"] + files, stderr=subprocess.PIPE, universal_newlines=True)
    dt = time.time() - t0
    print("-O " + om + " %-6s rc %d  %.2f s  %.0f columns/s  %.2e reads/s   %s" % (" ".join(opts) or "(BAQ)", p.returncode, dt, L / dt, nreads / dt, " | ".join(p.stderr.strip().splitlines()[-2:])[:400] if p.stderr.strip() else ""), flush=True)
