"""oracle/capmapq.c restates htslib's sam_cap_mapq (`mpileup -C INT`, mpileup.c:235-239).  No golden of the reference's tests
runs `mpileup -C`, so the restatement is PARITY UNPINNED; what is checked here is the closed form on hand-made reads:
    t = sum of mismatch qualities (each <= 33) - 4.343 ln C(len, mm) + clip_q / 5,  cap = (int)(sqrt((thres - t) / thres) thres + .499)
with len counting every aligned base twice when it is unambiguous and of quality >= 13 (the `++len` and `len += l` of the source)."""
import math

import numpy as np

from tests.helpers import mplpdrv as M, sam


def read(pos, cigar, seq, qual, mapq=60):
    r = sam.Read.__new__(sam.Read)
    r.pos, r.cigar, r.seq, r.qual, r.mapq, r.flag, r.l_qseq = pos, cigar, seq, np.array(qual, dtype=np.uint8), mapq, 0, len(seq)
    return r


def want(q_sum, mm, length, clip_q, thres):
    t = 1.0
    for i in range(mm):
        t *= length / (i + 1)
    t = q_sum - 4.343 * math.log(t) + clip_q / 5.0
    if t > thres:
        return -1
    t = max(t, 0.0)
    return int(math.sqrt((thres - t) / thres) * thres + .499)


REF = "ACGTACGTAGCTAGCTAGGATCGATCGATTTACGCGCGATATCGCGCTAGCTAGCATCGACTAGCTAGCTACGACGATCAGCATCGACT" * 3


def test_perfect_read_gets_the_threshold():
    r = read(10, [(40, "M")], REF[10:50], [30] * 40)
    assert M.cap_mapq_oracle(r, REF, 50) == 50 == want(0, 0, 80, 0, 50)


def test_mismatches_lower_the_cap_and_enough_of_them_drop_the_read():
    seq = list(REF[10:50])
    caps = []
    for nmm in range(0, 8):
        s = seq[:]
        for k in range(nmm):
            s[3 + 4 * k] = "A" if s[3 + 4 * k] != "A" else "C"
        r = read(10, [(40, "M")], "".join(s), [40] * 40)
        got = M.cap_mapq_oracle(r, REF, 50)
        assert got == want(33 * nmm, nmm, 80, 0, 50), nmm
        caps.append(got)
    assert caps[0] == 50 and caps[-1] == -1 and all(a >= b for a, b in zip(caps, caps[1:]))


def test_clips_low_qualities_and_ambiguous_bases():
    # 5 soft-clipped bases of quality 20, a mismatch of quality 10 (ignored: < 13), an N in the read, a hard clip
    seq = list(REF[20:60])
    seq[7] = "A" if seq[7] != "A" else "C"
    seq[12] = "N"
    seq[20] = "T" if seq[20] != "T" else "G"
    qual = [30] * 45
    qual[5 + 7] = 10
    r = read(20, [(3, "H"), (5, "S"), (40, "M")], "ACGTA" + "".join(seq), qual)
    # aligned bases counted once more when unambiguous and of quality >= 13: 40 + 38; one counted mismatch of quality 30
    assert M.cap_mapq_oracle(r, REF, 60) == want(30, 1, 78, 5 * 30 + 13 * 3, 60)


def test_default_threshold_and_the_end_of_the_reference():
    r = read(len(REF) - 20, [(40, "M")], REF[-20:] + "A" * 20, [30] * 40)
    # the walk stops where the reference ends: nothing of the operation that runs over it is added
    assert M.cap_mapq_oracle(r, REF, -1) == 40
