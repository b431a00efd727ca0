"""The read pool kept in HBM (bcfgpu_pool_upload -> bcfgpu_pool_baq -> bcfgpu_pool_cap_mapq -> bcfgpu_pool_keep ->
bcfgpu_pool_overlap_tweak -> bcfgpu_pool_pileup) against the same stages with host pointers between them (bcfgpu_baq,
bcfgpu_cap_mapq, bcfgpu_overlap_tweak, bcfgpu_pileup), which the other GPU tests pin to the oracle and the goldens: the
reference's SAM fixtures, raw reads in, every intermediate (new qualities, ZQ bytes, BAQ's ret, the caps, the lowered mapping
qualities) and the final tile compared."""
import copy
import ctypes as C
import os

import numpy as np
import pytest

from bcftools_amd import abi
from bcftools_amd.lib import check
from tests.helpers import sam, mplpdrv as M
from tests.test_gpu_pileup import assert_tiles_equal, device_pileup

pytestmark = pytest.mark.gpu


def _raw(golden_dir, files, fa, contig):
    G = os.path.join(golden_dir, "mpileup")
    sams = [sam.Sam(os.path.join(G, f)) for f in files]
    ref = sam.read_fasta(os.path.join(G, fa))
    prep = M.Prepared(sams, ref, contig, sam.MplpOpts(), baq=False, overlaps=False)
    return prep


def _tile_of(ctx, t, n_sites, S):
    from bcftools_amd import host
    off = np.zeros(n_sites * S + 1, np.uint32)
    ref16 = np.zeros(n_sites, np.int8)
    w = np.zeros(int(t.n_reads), np.uint32)
    e = np.zeros(int(t.n_reads), np.uint8)
    for dst, src in ((off, t.plp_off), (ref16, t.ref16), (w, t.rd), (e, t.epos)):
        if dst.nbytes:
            check(ctx.L.bcfgpu_memcpy_d2h(ctx.h, dst.ctypes.data, src, dst.nbytes))
    ctx.sync()
    return host.HostTile(S, ref16, off, w, e)


@pytest.mark.parametrize("files,fa,contig,beg,end,flag,packed", [
    (["mpileup.1.sam", "mpileup.2.sam", "mpileup.3.sam"], "mpileup.ref.fa", "17", 0, 700, 3, True),
    (["mpileup.1.sam", "mpileup.2.sam", "mpileup.3.sam"], "mpileup.ref.fa", "17", 100, 600, 7, False),
    (["indel-AD.1.sam"], "indel-AD.1.fa", "000000F", 0, 1200, 3, True),
])
def test_pool_chain_matches_the_host_pointer_chain(golden_dir, gpu_ctx_factory, files, fa, contig, beg, end, flag, packed):
    prep = _raw(golden_dir, files, fa, contig)
    S = len(prep.samples)
    by_sample = [[] for _ in range(S)]
    for rl in prep.files:
        for r, si in rl:
            by_sample[si].append(r)
    reads = [r for rl in by_sample for r in rl]
    smpl = np.array([si for si, rl in enumerate(by_sample) for _ in rl], np.int32)
    n = len(reads)
    refb = prep.refseq.encode()
    ctx = gpu_ctx_factory(abi.default_cfg(S, max_sites=1, max_reads=64))
    L = ctx.L
    thres = 50

    # ---- host pointers between the stages ----
    rd, d = M.pack_reads(reads)
    nb = len(d["qual"])
    qo, zo, ret = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8), np.zeros(n, np.int32)
    check(L.bcfgpu_baq(ctx.h, C.byref(rd), refb, len(prep.refseq), flag, qo.ctypes.data, zo.ctypes.data, ret.ctypes.data))
    assert (ret == 0).any()
    rd.qual = qo.ctypes.data
    cap = np.zeros(n, np.int32)
    check(L.bcfgpu_cap_mapq(ctx.h, C.byref(rd), refb, len(prep.refseq), thres, cap.ctypes.data))
    mapq = np.array([r.mapq for r in reads], np.uint8)
    mapq_a = np.where((cap >= 0) & (mapq > cap), cap, mapq).astype(np.uint8)
    keep = (cap >= 0).astype(np.uint8)
    keep[::17] = 0                                                  # and some dropped by a filter of the caller's
    # pairs among the kept reads, file by file as the iterator sees them
    index = {id(r): i for i, r in enumerate(reads)}
    pa, pb = [], []
    for rl in prep.files:
        for a, b in M.overlap_pairs([r for r, _ in rl if keep[index[id(r)]]]):
            pa.append(index[id(a)]); pb.append(index[id(b)])
    pa, pb = np.array(pa, np.int32), np.array(pb, np.int32)
    q2 = np.zeros(nb, np.uint8)
    check(L.bcfgpu_overlap_tweak(ctx.h, C.byref(rd), len(pa), pa.ctypes.data, pb.ctypes.data, q2.ctypes.data))
    if len(pa) == 0:
        q2 = qo.copy()
    # the kept reads as a pool of their own, with the qualities and mapping qualities the stages left
    kept_by_sample = []
    for si, rl in enumerate(by_sample):
        out = []
        for r in rl:
            i = index[id(r)]
            if not keep[i]:
                continue
            o = int(d["r_seq_off"][i])
            r2 = copy.copy(r)
            r2.qual = q2[o:o + r.l_qseq].astype(np.int32)
            r2.mapq = int(mapq_a[i])
            out.append(r2)
        kept_by_sample.append(out)
    want, want_n, want_indel, _ = device_pileup(ctx, kept_by_sample, prep.refseq, beg, end)

    # ---- the pool in HBM ----
    rd2, d2 = M.pack_reads(reads)
    pk = None
    if packed:
        pk = abi.Packed()
        seq4 = abi.pack_nibbles(d2["seq16"])
        pk.seq4, pk.n_bases, pk.n_cig = seq4.ctypes.data, nb, len(d2["cig"])
        rd2.seq16 = None
    check(L.bcfgpu_pool_upload(ctx.h, C.byref(rd2), C.byref(pk) if pk is not None else None, mapq.ctypes.data))
    ret_b = np.full(n, 99, np.int32)
    check(L.bcfgpu_pool_baq(ctx.h, refb, len(prep.refseq), flag, ret_b.ctypes.data))
    np.testing.assert_array_equal(ret_b, ret)
    q_b, z_b = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8)
    check(L.bcfgpu_pool_download(ctx.h, q_b.ctypes.data, z_b.ctypes.data, None))
    np.testing.assert_array_equal(q_b, qo)
    np.testing.assert_array_equal(z_b, zo)
    cap_b = np.zeros(n, np.int32)
    check(L.bcfgpu_pool_cap_mapq(ctx.h, refb, len(prep.refseq), thres, cap_b.ctypes.data))
    np.testing.assert_array_equal(cap_b, cap)
    m_b = np.zeros(n, np.uint8)
    check(L.bcfgpu_pool_download(ctx.h, None, None, m_b.ctypes.data))
    np.testing.assert_array_equal(m_b, mapq_a)
    check(L.bcfgpu_pool_keep(ctx.h, keep.ctypes.data))
    check(L.bcfgpu_pool_overlap_tweak(ctx.h, len(pa), pa.ctypes.data if len(pa) else None, pb.ctypes.data if len(pb) else None))
    check(L.bcfgpu_pool_download(ctx.h, q_b.ctypes.data, None, None))
    np.testing.assert_array_equal(q_b, q2)
    t = abi.Tile()
    n_sites = end - beg
    col_n, col_indel = np.zeros(n_sites, np.int32), np.zeros(n_sites, np.uint8)
    check(L.bcfgpu_pool_pileup(ctx.h, smpl.ctypes.data, None, beg, end, refb, len(prep.refseq), C.byref(t), col_n.ctypes.data, col_indel.ctypes.data))
    got = _tile_of(ctx, t, n_sites, S)
    assert_tiles_equal(got, want)
    np.testing.assert_array_equal(col_n, want_n)
    np.testing.assert_array_equal(col_indel, want_indel)
    assert len(want.rd) > 1000


def test_empty_pool_and_keep_reset(gpu_ctx_factory):
    """A pool without reads goes through every stage and gives an empty tile; bcfgpu_pool_keep(NULL) lets every read in again."""
    ctx = gpu_ctx_factory(abi.default_cfg(3, max_sites=1, max_reads=64))
    L = ctx.L
    rd = abi.Reads()
    check(L.bcfgpu_pool_upload(ctx.h, C.byref(rd), None, None))
    check(L.bcfgpu_pool_baq(ctx.h, b"ACGTACGT", 8, 3, None))
    check(L.bcfgpu_pool_overlap_tweak(ctx.h, 0, None, None))
    t = abi.Tile()
    col_n = np.ones(8, np.int32)
    check(L.bcfgpu_pool_pileup(ctx.h, None, None, 0, 8, b"ACGTACGT", 8, C.byref(t), col_n.ctypes.data, None))
    assert t.n_reads == 0 and t.n_sites == 8 and not col_n.any()
    # a small pool: one read per sample; dropping a read by the mask and letting it in again
    rng = np.random.default_rng(4)
    from tests.helpers import ovlfuzz
    by_sample = [[ovlfuzz.make_read(rng, 2 + s, 40)] for s in range(3)]
    for rl in by_sample:
        for r in rl:
            r.mapq, r.flag = 50, 0
    reads = [r for rl in by_sample for r in rl]
    rd2, d2 = M.pack_reads(reads)
    mapq = np.full(3, 50, np.uint8)
    smpl = np.arange(3, dtype=np.int32)
    refseq = "ACGT" * 30
    check(L.bcfgpu_pool_upload(ctx.h, C.byref(rd2), None, mapq.ctypes.data))
    full, n_full, _, _ = device_pileup(ctx, by_sample, refseq, 0, 80)
    check(L.bcfgpu_pool_upload(ctx.h, C.byref(rd2), None, mapq.ctypes.data))
    keep = np.array([1, 0, 1], np.uint8)
    check(L.bcfgpu_pool_keep(ctx.h, keep.ctypes.data))
    cn = np.zeros(80, np.int32)
    check(L.bcfgpu_pool_pileup(ctx.h, smpl.ctypes.data, None, 0, 80, refseq.encode(), len(refseq), C.byref(t), cn.ctypes.data, None))
    assert 0 < cn.sum() < n_full.sum()
    check(L.bcfgpu_pool_keep(ctx.h, None))
    check(L.bcfgpu_pool_pileup(ctx.h, smpl.ctypes.data, None, 0, 80, refseq.encode(), len(refseq), C.byref(t), cn.ctypes.data, None))
    np.testing.assert_array_equal(cn, n_full)
    assert_tiles_equal(_tile_of(ctx, t, 80, 3), full)


@pytest.mark.parametrize("seed,flag", [(5, 3), (6, 7), (7, 1)])
def test_pool_baq_matches_the_host_pointer_call_on_random_reads(gpu_ctx_factory, seed, flag):
    """bcfgpu_pool_baq (windows and bands on the device, the reference slice uploaded once) against bcfgpu_baq (windows on the
    host, a window copy per read), which tests/test_gpu_baq.py pins to the oracle: reads with every CIGAR operation, starting
    before the window can open and ending past the reference, unmapped reads, reads without qualities, long indels (the
    wide-band class)."""
    from tests.helpers import ovlfuzz
    rng = np.random.default_rng(seed)
    Lr = 400
    refseq = "".join("ACGTN"[i] for i in rng.choice(5, Lr, p=[0.25, 0.25, 0.24, 0.24, 0.02]))
    reads = []
    for k in range(300):
        pos = int(rng.choice([0, 1, 2, Lr - 30, Lr - 5])) if k % 7 == 0 else int(rng.integers(0, Lr - 20))
        r = ovlfuzz.make_read(rng, pos, int(rng.integers(1, 140)))
        r.flag = int(rng.choice([0, 16, 4, 0, 0]))
        r.mapq = 30
        if k % 11 == 0:
            r.qual = np.full(r.l_qseq, 255, np.int32)                # no qualities: left alone
        if k % 13 == 0 and r.l_qseq > 40:                             # one long deletion: a band wider than the register rows take
            a = r.l_qseq // 2
            r.cigar = [(a, "M"), (int(rng.integers(9, 25)), "D"), (r.l_qseq - a, "M")]
            r.bamcigar = np.array([n << 4 | "MIDNSHP=X".index(op) for n, op in r.cigar], dtype=np.uint32)
        reads.append(r)
    rd, d = M.pack_reads(reads)
    nb, n = len(d["qual"]), len(reads)
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
    L = ctx.L
    qo, zo, ret = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8), np.zeros(n, np.int32)
    check(L.bcfgpu_baq(ctx.h, C.byref(rd), refseq.encode(), Lr, flag, qo.ctypes.data, zo.ctypes.data, ret.ctypes.data))
    assert (ret == 0).sum() > 100 and (ret < 0).sum() > 20
    mapq = np.full(n, 30, np.uint8)
    check(L.bcfgpu_pool_upload(ctx.h, C.byref(rd), None, mapq.ctypes.data))
    ret_b = np.full(n, 99, np.int32)
    check(L.bcfgpu_pool_baq(ctx.h, refseq.encode(), Lr, flag, ret_b.ctypes.data))
    q_b, z_b = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8)
    check(L.bcfgpu_pool_download(ctx.h, q_b.ctypes.data, z_b.ctypes.data, None))
    np.testing.assert_array_equal(ret_b, ret)
    np.testing.assert_array_equal(q_b, qo)
    np.testing.assert_array_equal(z_b, zo)
    # a second BAQ over the pool's new qualities (mpileup -E redoes it): again what the host-pointer call gives on them
    rd.qual = qo.ctypes.data
    qo2, zo2 = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8)
    check(L.bcfgpu_baq(ctx.h, C.byref(rd), refseq.encode(), Lr, flag, qo2.ctypes.data, zo2.ctypes.data, ret.ctypes.data))
    check(L.bcfgpu_pool_baq(ctx.h, refseq.encode(), Lr, flag, None))
    check(L.bcfgpu_pool_download(ctx.h, q_b.ctypes.data, z_b.ctypes.data, None))
    np.testing.assert_array_equal(q_b, qo2)
    np.testing.assert_array_equal(z_b, zo2)


def test_pool_baq_at_scale_is_invariant_to_how_the_pool_is_cut(gpu_ctx_factory):
    """3e5 reads of 100 bases (3e7 bases: several scratch blocks of wavefronts): BAQ of the whole pool equals BAQ of its two
    halves uploaded one after the other -- a read's result depends on nothing but the read -- and two runs are identical."""
    from bcftools_amd import synth
    b = synth.indel_batch(77, 24, 420, depth=30.0)
    R = b["reads"]
    n = int(R["n_reads"])
    assert n > 2.5e5
    ctx = gpu_ctx_factory(abi.default_cfg(1, max_sites=1, max_reads=64))
    L = ctx.L
    mapq = np.full(n, 60, np.uint8)

    def run(lo, hi):
        rd = abi.Reads()
        rd.n_reads = hi - lo
        b0, c0 = int(R["r_seq_off"][lo]), int(R["r_cig_off"][lo])
        b1 = int(R["r_seq_off"][hi - 1] + R["r_lq"][hi - 1])
        arrs = dict(r_pos=R["r_pos"][lo:hi].copy(), r_lq=R["r_lq"][lo:hi].copy(), r_flag=R["r_flag"][lo:hi].copy(), r_ncig=R["r_ncig"][lo:hi].copy(),
                    r_cig_off=(R["r_cig_off"][lo:hi] - c0).astype(np.int32), r_seq_off=(R["r_seq_off"][lo:hi] - b0).astype(np.int32),
                    cig=R["cig"][c0:].copy(), seq16=R["seq16"][b0:b1].copy(), qual=R["qual"][b0:b1].copy())
        for k, v in arrs.items():
            setattr(rd, k, v.ctypes.data)
        check(L.bcfgpu_pool_upload(ctx.h, C.byref(rd), None, mapq[lo:hi].ctypes.data))
        ret = np.zeros(hi - lo, np.int32)
        check(L.bcfgpu_pool_baq(ctx.h, b["ref"], len(b["ref"]), 3, ret.ctypes.data))
        q, z = np.zeros(b1 - b0, np.uint8), np.zeros(b1 - b0, np.uint8)
        check(L.bcfgpu_pool_download(ctx.h, q.ctypes.data, z.ctypes.data, None))
        return q, z, ret
    q, z, ret = run(0, n)
    q_again, z_again, _ = run(0, n)
    assert q.tobytes() == q_again.tobytes() and z.tobytes() == z_again.tobytes()
    cut = n // 2 + 13
    qa, za, ra = run(0, cut)
    qb, zb, rb = run(cut, n)
    assert np.concatenate([qa, qb]).tobytes() == q.tobytes() and np.concatenate([za, zb]).tobytes() == z.tobytes()
    np.testing.assert_array_equal(np.concatenate([ra, rb]), ret)
    assert (ret == 0).mean() > 0.9 and (q != R["qual"][:len(q)]).mean() > 0.01      # BAQ did lower qualities


def test_pool_stages_need_a_pool(gpu_ctx_factory):
    ctx = gpu_ctx_factory(abi.default_cfg(2, max_sites=1, max_reads=64))
    t = abi.Tile()
    assert ctx.L.bcfgpu_pool_baq(ctx.h, b"ACGT", 4, 3, None) != 0
    assert ctx.L.bcfgpu_pool_pileup(ctx.h, None, None, 0, 4, b"ACGT", 4, C.byref(t), None, None) != 0
    assert b"bcfgpu_pool_upload" in ctx.L.bcfgpu_last_error()
