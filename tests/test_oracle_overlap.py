"""The mate-overlap quality tweak (htslib sam.c tweak_overlap_quality, switched on at mpileup.c:640): the C oracle
(oracle/overlap.c) against the first, Python restatement -- the one the reference's goldens mpileup.{1,2,4,5}.out were
first reproduced with (tests/test_oracle_golden_baq.py now runs the C oracle) -- on the overlapping mates of the
reference's SAM fixtures and on random pairs."""
import os

import numpy as np
import pytest

from tests.helpers import sam, mplpdrv as M, ovlfuzz


def _both(pairs):
    """returns (qualities by the Python restatement, qualities by the C oracle) for every read of the pairs"""
    reads = [r for ab in pairs for r in ab]
    q0 = [r.qual.copy() for r in reads]
    for a, b in pairs:
        M.tweak_overlap_quality(a, b)
    want = [r.qual.copy() for r in reads]
    for r, q in zip(reads, q0):
        r.qual = q.copy()
    rd, d = M.pack_reads(reads)
    import ctypes as C
    from bcftools_amd import abi
    from tests.helpers import orc
    L = orc.lib()
    L.orc_overlap_tweak.restype = C.c_int
    L.orc_overlap_tweak.argtypes = [C.POINTER(abi.Reads), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    pa = np.arange(0, len(reads), 2, dtype=np.int32)
    pb = pa + 1
    qo = d["qual"].copy()
    assert L.orc_overlap_tweak(C.byref(rd), len(pairs), pa.ctypes.data, pb.ctypes.data, qo.ctypes.data) == 0
    got = [qo[o:o + r.l_qseq].astype(np.int32) for r, o in zip(reads, d["r_seq_off"])]
    for r, q in zip(reads, q0):
        r.qual = q
    return want, got, q0


@pytest.mark.parametrize("samf", ["mpileup.1.sam", "mpileup.2.sam", "mpileup.4.sam"])
def test_c_oracle_matches_python_on_reference_mates(golden_dir, samf):
    s = sam.Sam(os.path.join(golden_dir, "mpileup", samf))
    reads = [r for r in s.reads if sam.keep_read(r, sam.MplpOpts())]
    pairs = M.overlap_pairs(reads)
    assert len(pairs) > 0
    want, got, q0 = _both(pairs)
    changed = 0
    for w, g, q in zip(want, got, q0):
        np.testing.assert_array_equal(g, w)
        changed += int((w != q).any())
    assert changed > 0


def test_c_oracle_matches_python_on_random_pairs():
    pairs = ovlfuzz.pairs(5, 400)
    want, got, q0 = _both(pairs)
    n_changed = 0
    for w, g, q in zip(want, got, q0):
        np.testing.assert_array_equal(g, w)
        n_changed += int((w != q).any())
    assert n_changed > 200
    assert max(int(w.max()) for w in want) == 200          # the cap of pooled qualities was reached
