"""Pins test16 (ccall.c:103-138: Fisher's exact test + three t-tests through htslib's kf_betai), which `call -m -a PV4`
uses (mcall.c:1668-1678): the INFO/PV4 values of the reference's golden test/mpileup.c.1.out (test.pl:298) against the
INFO/I16 of its input test/mpileup.c.vcf at the same records, and DP4 with them (MQ is not comparable: `call -c` prints the
root mean square, `call -m` the mean, mcall.c:1665 -- that one is pinned by the `call -m` goldens)."""
import ctypes as C
import os

import numpy as np

from tests.helpers import orc, vcf


def _records(golden_dir):
    g = os.path.join(golden_dir, "call")
    src = {(r.pos, "INDEL" in r.info): r for r in vcf.Vcf(os.path.join(g, "mpileup.c.vcf")).recs}
    out = [r for r in vcf.Vcf(os.path.join(g, "mpileup.c.1.out")).recs]
    return [(src[(r.pos, "INDEL" in r.info)], r) for r in out]


def run_test16(anno):
    L = orc.lib()
    L.orc_test16.restype = C.c_int
    L.orc_test16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    a = np.ascontiguousarray(anno, dtype=np.float32)
    p = np.zeros(4, dtype=np.float64)
    t = C.c_int(0)
    rc = L.orc_test16(a.ctypes.data, p.ctypes.data, C.byref(t))
    return rc, p, t.value


def test_pv4_of_reference_golden(golden_dir):
    n_pv4 = n_nontrivial = 0
    for rin, rout in _records(golden_dir):
        anno = rin.info_floats("I16")
        rc, p, tested = run_test16(anno)
        assert rc == 0
        assert rout.info_ints("DP4") == [int(x) for x in anno[:4]]
        if "PV4" in rout.info:
            assert tested
            want = rout.info_floats("PV4")
            for a, b in zip(p, want):
                assert abs(np.float32(a) - b) <= 5e-6 * max(abs(b), 1e-30) + 1e-12, (rout.pos, p, want)
                n_nontrivial += int(b != 1.0)
            n_pv4 += 1
        else:
            assert not tested
    assert n_pv4 == 11 and n_nontrivial >= 15
