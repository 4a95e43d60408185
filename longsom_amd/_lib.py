"""ctypes binding of liblongsom_hip.so (include/longsom_hip.h).

The library is the product's only compute path: if it is missing or cannot be loaded the import of
any compute entry point fails loudly — there is no CPU fallback here (the CPU restatement lives in
/oracle and is test infrastructure only).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSG_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "liblongsom_hip.so")   # override: tuning builds only

ROW_WORDS = 42
MAX_CELLTYPES = 4
CALL_MAX_ALT = 4

SYM_NAMES = ["A", "C", "T", "G", "I", "D", "N", "O"]
SYM_NA = 15

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)


class Reads(C.Structure):
    _fields_ = [
        ("n_reads", C.c_int64), ("n_segs", C.c_int64), ("n_events", C.c_int64),
        ("read_tid", C.c_void_p), ("read_pos", C.c_void_p), ("read_flag", C.c_void_p),
        ("read_mapq", C.c_void_p), ("read_cb", C.c_void_p),
        ("seg_read", C.c_void_p), ("seg_start", C.c_void_p), ("seg_len", C.c_void_p),
        ("seg_ev_off", C.c_void_p), ("events", C.c_void_p),
        ("on_device", C.c_int32),
    ]


class CountParams(C.Structure):
    _fields_ = [
        ("min_bq", C.c_int32), ("min_mq", C.c_int32), ("min_dp", C.c_int32), ("min_cc", C.c_int32),
        ("flag_exclude", C.c_uint32), ("ignore_orphans", C.c_int32), ("max_depth", C.c_int32),
    ]

    @classmethod
    def longsom_defaults(cls, **kw):
        """Flags as LongSom's rules run BaseCellCounter (R:SNVCalling.smk:52-59, BaseCellCounter.py:331-339; max_depth :191)."""
        p = cls(min_bq=20, min_mq=60, min_dp=5, min_cc=5, flag_exclude=0xF04, ignore_orphans=1, max_depth=200000)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class GenotypeParams(C.Structure):
    _fields_ = [("min_bq", C.c_int32), ("min_mq", C.c_int32), ("flag_exclude", C.c_uint32), ("ignore_orphans", C.c_int32),
                ("alt_only", C.c_int32), ("strict_cb", C.c_int32)]

    @classmethod
    def longsom_defaults(cls, **kw):
        """HCCVSingleCellGenotype.py:327-329 (min_bq 30) with the rule's --min_mq (R:CellTypeReannotation.smk:320,344)."""
        p = cls(min_bq=30, min_mq=60, flag_exclude=0xF04, ignore_orphans=1, alt_only=0, strict_cb=1)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class CallParams(C.Structure):
    _fields_ = [
        ("alpha1", C.c_double), ("beta1", C.c_double), ("alpha2", C.c_double), ("beta2", C.c_double),
        ("min_cov", C.c_int32), ("min_cells", C.c_int32), ("min_ac_cells", C.c_int32),
        ("min_ac_reads", C.c_int32), ("max_cell_types", C.c_int32), ("min_cell_types", C.c_int32),
    ]

    @classmethod
    def longsom_defaults(cls, **kw):
        """config/config.yaml:77-90 + BaseCellCalling.step1.py:592-603."""
        p = cls(alpha1=0.21356677091082193, beta1=104.95163748636298,
                alpha2=0.2474528917555431, beta2=162.03696139428595,
                min_cov=5, min_cells=5, min_ac_cells=2, min_ac_reads=3, max_cell_types=1, min_cell_types=2)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class Call(C.Structure):
    _fields_ = [
        ("key", C.c_int64),
        ("ref", C.c_uint8), ("present", C.c_uint8), ("considered", C.c_uint8), ("has_cand", C.c_uint8),
        ("n_alt", C.c_uint8 * MAX_CELLTYPES),
        ("alt", (C.c_uint8 * CALL_MAX_ALT) * MAX_CELLTYPES),
        ("ct_filter", C.c_uint8 * MAX_CELLTYPES),
        ("alt_bc", (C.c_uint32 * CALL_MAX_ALT) * MAX_CELLTYPES),
        ("alt_cc", (C.c_uint32 * CALL_MAX_ALT) * MAX_CELLTYPES),
        ("p_bc", (C.c_int32 * CALL_MAX_ALT) * MAX_CELLTYPES),
        ("p_cc", (C.c_int32 * CALL_MAX_ALT) * MAX_CELLTYPES),
        ("site_filter", C.c_uint32),
        ("cell_types_min", C.c_int32),
        ("sum_alts_bc", C.c_int32), ("sum_dp", C.c_int32), ("sum_alts_cc", C.c_int32), ("sum_nc", C.c_int32),
        ("noise_p_bc", C.c_int32), ("noise_p_cc", C.c_int32),
        ("up_ctx", C.c_uint8 * 5), ("down_ctx", C.c_uint8 * 5),
        ("pad", C.c_uint8 * 2),
    ]


class BamInfo(C.Structure):
    _fields_ = [
        ("total_reads", C.c_int64), ("pass_reads", C.c_int64), ("cb_not_found", C.c_int64), ("cb_not_matched", C.c_int64), ("mapq_filtered", C.c_int64),
        ("n_records", C.c_int64), ("n_blocks", C.c_int64), ("n_ubytes", C.c_int64),
        ("ms_h2d", C.c_float), ("ms_inflate", C.c_float), ("ms_chain", C.c_float), ("ms_decode", C.c_float), ("ms_store", C.c_float), ("ms_total", C.c_float),
        ("chain_rounds", C.c_int32), ("pad_", C.c_int32),
        ("last_key", C.c_int64),
    ]


class CountStats(C.Structure):
    _fields_ = [
        ("n_reads_admitted", C.c_int64), ("n_segs_admitted", C.c_int64), ("n_events_admitted", C.c_int64),
        ("n_entries", C.c_int64), ("n_units", C.c_int64), ("n_deep_units", C.c_int64),
        ("n_events_wave", C.c_int64), ("n_events_deep", C.c_int64), ("n_rows_wave", C.c_int64), ("n_rows_deep", C.c_int64),
        ("ms_bin", C.c_float), ("ms_deep", C.c_float), ("ms_wave", C.c_float), ("ms_total", C.c_float),
        ("ms_walk", C.c_float), ("pad_", C.c_float),
        ("rows_by_kernel", C.c_int64 * 4), ("events_by_kernel", C.c_int64 * 4),
    ]


# every symbol include/longsom_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "lsg_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "lsg_destroy": (None, [C.c_void_p]),
    "lsg_last_error": (C.c_char_p, []),
    "lsg_version": (C.c_char_p, []),
    "lsg_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lsg_synchronize": (C.c_int, [C.c_void_p]),
    "lsg_set_contigs": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "lsg_load_reference": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32]),
    "lsg_set_barcodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "lsg_load_reads": (C.c_int, [C.c_void_p, C.POINTER(Reads)]),
    "lsg_set_region": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int64]),
    "lsg_set_count_at_load": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lsg_set_store_policy": (C.c_int, [C.c_void_p, C.c_int32]),
    "lsg_set_keep_unlisted": (C.c_int, [C.c_void_p, C.c_int32]),
    "lsg_set_pileup_window": (C.c_int, [C.c_void_p, C.c_int32]),
    "lsg_synth_reference": (C.c_int, [C.c_void_p, C.c_uint64]),
    "lsg_synth_reads": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lsg_get_reads_shape": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lsg_copy_reads_to_host": (C.c_int, [C.c_void_p, C.POINTER(Reads)]),
    "lsg_copy_reference_to_host": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "lsg_pileup_count": (C.c_int, [C.c_void_p, C.POINTER(CountParams), C.c_void_p, C.c_void_p]),
    "lsg_fetch_counts": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "lsg_load_counts": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lsg_call_step1": (C.c_int, [C.c_void_p, C.POINTER(CallParams), C.c_void_p, C.c_void_p]),
    "lsg_fetch_calls": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "lsg_export_calls": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "lsg_set_table_names": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_char_p]),
    "lsg_format_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "lsg_copy_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "lsg_append_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p]),
    "lsg_free_table": (C.c_int, [C.c_void_p, C.c_int32]),
    "lsg_step2_summary": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "lsg_load_posset": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32]),
    "lsg_probe_posset": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]),
    "lsg_genotype_cells": (C.c_int, [C.c_void_p, C.POINTER(GenotypeParams), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "lsg_genotype_cells_grouped": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "lsg_betabinom_sf4": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]),
    "lsg_betabinom_sf": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "lsg_max_live_reads": (C.c_int64, [C.c_void_p]),
    "lsg_get_count_stats": (C.c_int, [C.c_void_p, C.POINTER(CountStats)]),
    "lsg_get_layout_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "lsg_get_build_times": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lsg_get_store_shape": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lsg_set_keep_reads": (C.c_int, [C.c_void_p, C.c_int32]),
    "lsg_unload_reads": (C.c_int, [C.c_void_p]),
    "lsg_load_bam": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(BamInfo), C.c_void_p, C.c_void_p, C.c_int64]),
    "lsg_load_bam_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.POINTER(BamInfo), C.c_void_p, C.c_void_p, C.c_int64]),
    "lsg_set_load_filter": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint32, C.c_int32]),
    "lsg_set_events_layout": (C.c_int, [C.c_void_p, C.c_int32]),
    "lsg_max_live_reads_all": (C.c_int64, [C.c_void_p]),
    "lsg_max_live_reads_exact": (C.c_int64, [C.c_void_p]),
    "lsg_synth_generate": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Reads)]),
}

_lib = None


class LibraryMissing(RuntimeError):
    pass


def load():
    """Load liblongsom_hip.so; raises LibraryMissing (never falls back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C longsom_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise LibraryMissing(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class LsgError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().lsg_last_error().decode("utf-8", "replace")
        raise LsgError(f"{what} failed (rc={rc}): {msg}")
