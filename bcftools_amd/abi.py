"""ctypes mirror of include/bcfgpu.h (structs, constants, prototypes).

Plain data definitions only; no computation lives here.  Struct field order and
types must match the header exactly (checked by tests/test_abi.py against the
sizes the C library reports).
"""
import ctypes as C

# error codes
OK, E_ARG, E_NOMEM, E_HIP, E_DEPTH, E_NODEV, E_RANGE = 0, -1, -2, -3, -4, -5, -6

# B2B_* (bam2bcf.h:46-62)
FMT_DP, FMT_SP, FMT_DV, FMT_DP4, FMT_DPR, INFO_DPR = 1 << 0, 1 << 1, 1 << 2, 1 << 3, 1 << 4, 1 << 5
FMT_AD, FMT_ADF, FMT_ADR, INFO_AD, INFO_ADF, INFO_ADR = 1 << 6, 1 << 7, 1 << 8, 1 << 9, 1 << 10, 1 << 11
INFO_SCR, FMT_SCR, INFO_VDB, INFO_RPB, FMT_QS = 1 << 12, 1 << 13, 1 << 14, 1 << 15, 1 << 16
# CALL_* (call.h:32-39)
CALL_KEEPALT, CALL_VARONLY, CALL_FMT_PV4, CALL_FMT_GQ, CALL_FMT_GP = 1, 1 << 1, 1 << 5, 1 << 6, 1 << 7

MAX_ALLELES, MAX_PL, NPOS, NQUAL, MAX_DEPTH = 5, 15, 100, 60, 255
INT32_MISSING = -2147483648
INT32_VECTOR_END = -2147483647
GT_MISSING, GT_VECTOR_END = -1, -2

RD_REV, RD_SCLIP, RD_DEL, RD_SKIP = 1 << 20, 1 << 21, 1 << 22, 1 << 23


class Cfg(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("n_smpl", C.c_int32), ("max_sites", C.c_int32),
        ("max_reads", C.c_uint64),
        ("min_baseQ", C.c_int32), ("capQ", C.c_int32), ("errmod_theta", C.c_double),
        ("fmt_flag", C.c_int32),
        ("call_theta", C.c_double), ("call_flag", C.c_int32), ("output_tags", C.c_int32),
        ("n_grp", C.c_int32), ("grp_tag_is_qs", C.c_int32), ("ploidy_max", C.c_int32),
    ]


class Tile(C.Structure):
    _fields_ = [
        ("n_sites", C.c_int32), ("is_indel", C.c_int32), ("n_reads", C.c_uint64),
        ("ref16", C.c_void_p), ("plp_off", C.c_void_p), ("rd", C.c_void_p),
        ("epos", C.c_void_p), ("aux", C.c_void_p),
    ]


class Site(C.Structure):
    _fields_ = [
        ("a", C.c_int32 * 5), ("n_alleles", C.c_int32), ("unseen", C.c_int32),
        ("ori_ref", C.c_int32), ("shift", C.c_int32), ("ret", C.c_int32),
        ("depth", C.c_uint32), ("ori_depth", C.c_uint32), ("mq0", C.c_uint32),
        ("qsum", C.c_float * 5),
        ("vdb", C.c_float), ("mwu_pos", C.c_float), ("mwu_mq", C.c_float),
        ("mwu_bq", C.c_float), ("mwu_mqs", C.c_float), ("seg_bias", C.c_float),
        ("adf_tot", C.c_int32 * 5), ("adr_tot", C.c_int32 * 5),
        ("scr_tot", C.c_int32), ("pad", C.c_int32),
        ("anno", C.c_double * 16),
    ]


class MplpOut(C.Structure):
    _fields_ = [
        ("site", C.c_void_p), ("pl", C.c_void_p), ("dp4", C.c_void_p),
        ("adf", C.c_void_p), ("adr", C.c_void_p), ("qs", C.c_void_p), ("scr", C.c_void_p), ("sp", C.c_void_p),
    ]


class CallIn(C.Structure):
    _fields_ = [
        ("n_sites", C.c_int32), ("n_gt_max", C.c_int32), ("n_al_max", C.c_int32), ("reserved", C.c_int32),
        ("nals", C.c_void_p), ("unseen", C.c_void_p), ("pl", C.c_void_p), ("qs", C.c_void_p),
        ("ad", C.c_void_p), ("ploidy", C.c_void_p), ("grp", C.c_void_p),
        ("prior_an", C.c_void_p), ("prior_ac", C.c_void_p), ("i16", C.c_void_p),
    ]


class CallSite(C.Structure):
    _fields_ = [
        ("ret", C.c_int32), ("nals_new", C.c_int32), ("als_new", C.c_int32),
        ("als_map", C.c_int32 * 5), ("ac", C.c_int32 * 5), ("an", C.c_int32),
        ("qual_missing", C.c_int32), ("qual", C.c_float), ("pl_dropped", C.c_int32),
        ("has_i16", C.c_int32), ("dp4", C.c_int32 * 4), ("mq", C.c_int32), ("pv4_tested", C.c_int32), ("pv4", C.c_float * 4),
    ]


class CallOut(C.Structure):
    _fields_ = [
        ("site", C.c_void_p), ("gt", C.c_void_p), ("pl", C.c_void_p),
        ("gq", C.c_void_p), ("gp", C.c_void_p),
    ]


class Reads(C.Structure):
    _fields_ = [("n_reads", C.c_int32)] + [(k, C.c_void_p) for k in
                ("r_pos", "r_lq", "r_flag", "r_ncig", "r_cig_off", "r_seq_off", "cig", "seq16", "qual", "zq", "r_has_zq")]


class Packed(C.Structure):
    """bcfgpu_packed: the pool as BAM records hold it (4-bit bases, optional 4-bit palette qualities)."""
    _fields_ = [("seq4", C.c_void_p), ("qual4", C.c_void_p), ("palette", C.c_uint8 * 16), ("n_bases", C.c_int64), ("n_cig", C.c_int64),
                ("smpl_off", C.c_void_p), ("qual_bits", C.c_int32), ("recs", C.c_void_p)]


def pack_nibbles(a):
    """Two 4-bit values per byte, the even index in the high nibble (bam_get_seq's order); odd lengths padded with 0."""
    import numpy as np
    a = np.asarray(a, np.uint8)
    if a.size & 1:
        a = np.concatenate([a, np.zeros(1, np.uint8)])
    return ((a[0::2] << 4) | (a[1::2] & 15)).astype(np.uint8)


READ12 = [("pos", "<i4"), ("lq", "<u2"), ("ncig", "u1"), ("flag8", "u1"), ("mapq", "u1"), ("pad", "u1", 3)]     # bcfgpu_read12


def read12(r_pos, r_lq, r_ncig, r_flag, r_mapq):
    """The per-read arrays as bcfgpu_read12 records (numpy structured array, 12 bytes a read)."""
    import numpy as np
    rec = np.zeros(len(r_pos), dtype=READ12)
    rec["pos"], rec["lq"], rec["ncig"], rec["mapq"] = r_pos, r_lq, r_ncig, r_mapq
    rec["flag8"] = ((np.asarray(r_flag) & 16) != 0) * 1 + ((np.asarray(r_flag) & 4) != 0) * 2
    return rec


def pack_crumbs(a):
    """Four 2-bit values per byte, index 0 in the two highest bits; lengths padded to a multiple of four with 0."""
    import numpy as np
    a = np.asarray(a, np.uint8)
    if a.size & 3:
        a = np.concatenate([a, np.zeros(4 - (a.size & 3), np.uint8)])
    return ((a[0::4] << 6) | ((a[1::4] & 3) << 4) | ((a[2::4] & 3) << 2) | (a[3::4] & 3)).astype(np.uint8)


class IndelIn(C.Structure):
    _fields_ = [("n_sites", C.c_int32), ("n_smpl", C.c_int32), ("pos", C.c_void_p), ("smpl_off", C.c_void_p),
                ("p_read", C.c_void_p), ("p_qpos", C.c_void_p), ("p_indel", C.c_void_p), ("ref", C.c_char_p),
                ("openQ", C.c_int32), ("extQ", C.c_int32), ("tandemQ", C.c_int32), ("min_support", C.c_int32),
                ("per_sample_flt", C.c_int32), ("min_frac", C.c_double)]


class IndelOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("ret", "p_aux", "indel_types", "inscns", "maxins", "indelreg", "max_support", "max_frac")]


class GvcfBlock(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("first_site", "last_site", "start_pos", "end1", "min_dp", "range")]


class GvcfIn(C.Structure):
    _fields_ = [("n_sites", C.c_int32), ("n_range", C.c_int32)] + [(k, C.c_void_p) for k in
                ("dp_range", "pos", "rid", "brk", "site", "pl", "dp4", "ref_only", "dp", "end")]


class GvcfOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("blk", "min_dp", "block", "dp", "pl")]


class GapStats(C.Structure):
    _fields_ = [("n_jobs", C.c_uint64), ("n_passes", C.c_uint64), ("dp_cells", C.c_uint64),
                ("kernel_ms", C.c_float), ("prepare_ms", C.c_float), ("finalize_ms", C.c_float), ("total_ms", C.c_float),
                ("n_wide", C.c_uint64), ("n_scratch", C.c_uint64), ("band_jobs", C.c_uint32 * 8)]


class Timing(C.Structure):
    _fields_ = [("glfgen_ms", C.c_float), ("combine_ms", C.c_float),
                ("mcall_ms", C.c_float), ("total_ms", C.c_float)]


# every symbol include/bcfgpu.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "bcfgpu_create": (C.c_int, [C.POINTER(Cfg), C.POINTER(C.c_void_p)]),
    "bcfgpu_destroy": (None, [C.c_void_p]),
    "bcfgpu_last_error": (C.c_char_p, []),
    "bcfgpu_device_count": (C.c_int, []),
    "bcfgpu_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "bcfgpu_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bcfgpu_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bcfgpu_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bcfgpu_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "bcfgpu_sync": (C.c_int, [C.c_void_p]),
    "bcfgpu_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "bcfgpu_host_free": (C.c_int, [C.c_void_p]),
    "bcfgpu_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bcfgpu_pack_read": (None, [C.c_int] * 9 + [C.c_void_p, C.c_int, C.c_int,
                                                C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]),
    "bcfgpu_mpileup": (C.c_int, [C.c_void_p, C.POINTER(Tile), C.POINTER(MplpOut)]),
    "bcfgpu_mcall": (C.c_int, [C.c_void_p, C.POINTER(CallIn), C.POINTER(CallOut)]),
    "bcfgpu_gap_prep": (C.c_int, [C.c_void_p, C.POINTER(Reads), C.POINTER(IndelIn), C.POINTER(IndelOut), C.c_int]),
    "bcfgpu_gap_prep_tile": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(Reads), C.POINTER(IndelIn), C.POINTER(IndelOut), C.c_int,
                                       C.POINTER(Tile)]),
    "bcfgpu_baq": (C.c_int, [C.c_void_p, C.POINTER(Reads), C.c_char_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bcfgpu_cap_mapq": (C.c_int, [C.c_void_p, C.POINTER(Reads), C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]),
    "bcfgpu_overlap_tweak": (C.c_int, [C.c_void_p, C.POINTER(Reads), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bcfgpu_pileup_indel_tile": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(Tile)]),
    "bcfgpu_pileup_entries": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "bcfgpu_pileup": (C.c_int, [C.c_void_p, C.POINTER(Reads), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32,
                                C.POINTER(Tile), C.c_void_p, C.c_void_p]),
    "bcfgpu_pileup_packed": (C.c_int, [C.c_void_p, C.POINTER(Reads), C.POINTER(Packed), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_char_p,
                                       C.c_int32, C.POINTER(Tile), C.c_void_p, C.c_void_p]),
    "bcfgpu_pool_upload": (C.c_int, [C.c_void_p, C.POINTER(Reads), C.POINTER(Packed), C.c_void_p]),
    "bcfgpu_pool_baq": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int, C.c_void_p]),
    "bcfgpu_pool_cap_mapq": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]),
    "bcfgpu_pool_keep": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bcfgpu_pool_overlap_tweak": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "bcfgpu_pool_pileup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(Tile),
                                     C.c_void_p, C.c_void_p]),
    "bcfgpu_pool_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bcfgpu_gvcf_blocks": (C.c_int, [C.c_void_p, C.POINTER(GvcfIn), C.POINTER(GvcfOut), C.POINTER(C.c_int32)]),
    "bcfgpu_gap_prep_stats": (C.c_int, [C.c_void_p, C.POINTER(GapStats)]),
    "bcfgpu_pipeline": (C.c_int, [C.c_void_p, C.POINTER(Tile), C.c_void_p, C.c_void_p,
                                  C.POINTER(MplpOut), C.POINTER(CallOut)]),
    "bcfgpu_mplp_out_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "bcfgpu_truncated_cells": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "bcfgpu_errmod_plan": (C.c_int, [C.c_void_p, C.POINTER(Tile), C.POINTER(Tile), C.c_void_p, C.c_void_p]),
    "bcfgpu_errmod_plan_visit": (C.c_int, [C.c_void_p, C.POINTER(Tile), C.c_void_p, C.POINTER(Tile), C.c_void_p, C.c_void_p]),
    "bcfgpu_errmod_seed": (C.c_int, [C.c_void_p, C.c_uint64]),
    "bcfgpu_errmod_state": (C.c_uint64, [C.c_void_p]),
    "bcfgpu_depth_cap": (C.c_int, [C.POINTER(Reads), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "bcfgpu_depth_cap_new": (C.c_void_p, [C.c_int32, C.c_int32]),
    "bcfgpu_depth_cap_push": (C.c_int, [C.c_void_p, C.POINTER(Reads), C.c_void_p, C.c_void_p]),
    "bcfgpu_depth_cap_reset": (None, [C.c_void_p]),
    "bcfgpu_depth_cap_free": (None, [C.c_void_p]),
    "bcfgpu_compact_calls": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(CallOut), C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "bcfgpu_compact_calls_async": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(CallOut), C.c_int32, C.c_int32, C.c_void_p,
                                             C.c_uint64, C.c_void_p]),
    "bcfgpu_compact_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "bcfgpu_comm_init_all": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]),
    "bcfgpu_comm_destroy": (None, [C.c_void_p]),
    "bcfgpu_gather_bytes": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    "bcfgpu_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "bcfgpu_timing_get": (C.c_int, [C.c_void_p, C.POINTER(Timing)]),
    "bcfgpu_abi_sizes": (None, [C.POINTER(C.c_int32)]),
}


def default_cfg(n_smpl, max_sites=0, max_reads=0, device=0, fmt_flag=INFO_VDB | INFO_RPB,
                min_baseQ=13, call_theta=1.1e-3, call_flag=0, output_tags=0, n_grp=1):
    """Defaults of `bcftools mpileup` (mpileup.c:937-950) and `bcftools call -m` (vcfcall.c:931-943)."""
    c = Cfg()
    c.device, c.n_smpl, c.max_sites, c.max_reads = device, n_smpl, max_sites, max_reads
    c.min_baseQ, c.capQ, c.errmod_theta, c.fmt_flag = min_baseQ, 60, 0.0, fmt_flag
    c.call_theta, c.call_flag, c.output_tags = call_theta, call_flag, output_tags
    c.n_grp, c.grp_tag_is_qs, c.ploidy_max = n_grp, 0, 2
    return c
