"""bcfgpu_depth_cap (host helper of the C-ABI): the per-file depth cap of htslib's pileup iterator (mpileup -d,
mpileup.c:646 bam_mplp_set_maxcnt) against a direct replay of bam_plp_push / bam_plp_next's buffer bookkeeping."""
import ctypes as C

import numpy as np
import pytest

from bcftools_amd import abi, lib


def _replay(pos, end, smpl, n_smpl, maxcnt):
    """htslib sam.c, one iterator per sample: bam_plp_auto pushes a read when no column can be produced (iter->pos has
    caught up with max_pos); bam_plp_push drops it if it starts at iter->pos while more than maxcnt nodes are allocated
    (the buffered reads plus the list's tail node); bam_plp_next releases a read at the first column it does not cover."""
    keep = np.zeros(len(pos), np.uint8)
    for s in range(n_smpl):
        idx = np.nonzero(smpl == s)[0]
        buf = []                       # ends of the buffered reads
        it_pos, max_pos = 0, -1
        for r in idx:
            # columns < max_pos have been handed out: reads ending at or before the last of them are gone
            if max_pos > it_pos:
                buf = [e for e in buf if e > max_pos - 1]
                it_pos = max_pos
            if it_pos == pos[r] and len(buf) + 1 > maxcnt:
                continue
            keep[r] = 1
            if end[r] > it_pos:        # the tail node is linked only then (a read without reference bases: not always)
                buf.append(end[r])
            max_pos = pos[r]
    return keep


@pytest.mark.parametrize("seed,n_smpl,n_reads,maxcnt", [(1, 1, 4000, 50), (2, 3, 6000, 20), (3, 5, 3000, 250), (4, 2, 500, 1), (5, 4, 2000, 0)])
def test_depth_cap_matches_iterator_replay(seed, n_smpl, n_reads, maxcnt):
    rng = np.random.default_rng(seed)
    smpl = np.sort(rng.integers(0, n_smpl, n_reads)).astype(np.int32)
    pos = np.zeros(n_reads, np.int32)
    for s in range(n_smpl):
        m = smpl == s
        # pile-ups: many reads share start positions
        pos[m] = np.sort(rng.integers(0, 60, m.sum()) * rng.integers(1, 4)).astype(np.int32)
    lens = rng.integers(1, 120, n_reads)
    lens[rng.random(n_reads) < 0.05] = 0                       # reads without a reference base (all clipped / inserted)
    dele = rng.integers(0, 30, n_reads) * (rng.random(n_reads) < 0.2)
    cig, coff = [], []
    for i in range(n_reads):
        coff.append(len(cig))
        a = int(lens[i])
        if dele[i] and a > 2:
            cig += [(a // 2) << 4 | 0, int(dele[i]) << 4 | 2, (a - a // 2) << 4 | 0]
        elif a == 0:
            cig += [7 << 4 | 4, 9 << 4 | 1]
        else:
            cig += [5 << 4 | 4, a << 4 | 0]
    cig = np.array(cig, np.uint32)
    coff = np.array(coff, np.int32)
    ncig = np.diff(np.r_[coff, len(cig)]).astype(np.int32)
    end = pos + lens + np.where((dele > 0) & (lens > 2), dele, 0)
    rd = abi.Reads()
    rd.n_reads = n_reads
    rd.r_pos, rd.r_ncig, rd.r_cig_off, rd.cig = pos.ctypes.data, ncig.ctypes.data, coff.ctypes.data, cig.ctypes.data
    keep = np.zeros(n_reads, np.uint8)
    L = lib.load()
    assert L.bcfgpu_depth_cap(C.byref(rd), smpl.ctypes.data, n_smpl, maxcnt, keep.ctypes.data) == 0
    if maxcnt <= 0:
        assert keep.all()
        return
    want = _replay(pos, end, smpl, n_smpl, maxcnt)
    np.testing.assert_array_equal(keep, want)
    assert 0 < keep.sum() < n_reads or maxcnt >= 250
    # the first read of every start position is always kept
    for s in range(n_smpl):
        m = np.nonzero(smpl == s)[0]
        first = m[np.r_[True, np.diff(pos[m]) != 0]]
        assert keep[first].all()
    # the same reads pushed through the iterator's state in batches (a region streamed in tiles, host/bcfgpu_sam.c): per
    # sample in position order, batch boundaries at arbitrary positions -- reads that start at a boundary position land on either side
    st = L.bcfgpu_depth_cap_new(n_smpl, maxcnt)
    assert st
    keep2 = np.zeros(n_reads, np.uint8)
    cuts = np.r_[-1, np.sort(rng.integers(0, 200, 7)), 10 ** 9]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        idx = np.nonzero((pos > lo) & (pos <= hi))[0]                 # the batch: every sample's reads of this position range, samples in order
        if not len(idx):
            continue
        b = abi.Reads()
        bp, bn, bo, bs = (np.ascontiguousarray(x[idx]) for x in (pos, ncig, coff, smpl))
        b.n_reads = len(idx)
        b.r_pos, b.r_ncig, b.r_cig_off, b.cig = bp.ctypes.data, bn.ctypes.data, bo.ctypes.data, cig.ctypes.data
        kb = np.zeros(len(idx), np.uint8)
        assert L.bcfgpu_depth_cap_push(st, C.byref(b), bs.ctypes.data, kb.ctypes.data) == 0
        keep2[idx] = kb
    np.testing.assert_array_equal(keep2, keep)
    L.bcfgpu_depth_cap_reset(st)
    kb = np.zeros(n_reads, np.uint8)
    assert L.bcfgpu_depth_cap_push(st, C.byref(rd), smpl.ctypes.data, kb.ctypes.data) == 0      # after a reset: a new region, the whole of it at once
    np.testing.assert_array_equal(kb, keep)
    L.bcfgpu_depth_cap_free(st)


def test_depth_cap_rejects_unsorted_reads():
    pos = np.array([10, 5], np.int32)
    z = np.zeros(2, np.int32)
    cig = np.array([10 << 4, 10 << 4], np.uint32)
    one = np.ones(2, np.int32)
    coff = np.array([0, 1], np.int32)
    rd = abi.Reads()
    rd.n_reads = 2
    rd.r_pos, rd.r_ncig, rd.r_cig_off, rd.cig = pos.ctypes.data, one.ctypes.data, coff.ctypes.data, cig.ctypes.data
    keep = np.zeros(2, np.uint8)
    assert lib.load().bcfgpu_depth_cap(C.byref(rd), z.ctypes.data, 1, 5, keep.ctypes.data) == abi.E_ARG
