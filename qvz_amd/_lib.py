"""Loader for the in-tree native library (qvz_amd/libqvzhip.so, built by `make` / __graft_entry__.build()).

There is NO fallback: if the HIP library is missing or fails to load, importing the kernels raises. The
Python layer is plumbing only (device memory via torch, stream handles, torch.distributed); every computation
on the encode path runs in the hand-written HIP kernels behind the C-ABI of include/qvz_hip.h.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libqvzhip.so")

u8p, u16p, u32p, u64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p   # device pointers travel as integers
f64p = C.c_void_p

# name -> (restype, argtypes); kept in one table so tests can check it against include/qvz_hip.h
SIGNATURES = {
    "qvz_hip_last_error": (C.c_char_p, []),
    "qvz_hip_set_error": (C.c_int, [C.c_int, C.c_char_p]),
    "qvz_hip_ld": (C.c_uint64, [C.c_uint64]),
    "qvz_hip_device_count": (C.c_int, []),
    "qvz_hip_transpose_text": (C.c_int, [u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint64, u32p, C.c_void_p]),
    "qvz_hip_untranspose_text": (C.c_int, [u8p, C.c_uint64, C.c_uint32, C.c_uint64, u8p, C.c_void_p]),
    "qvz_hip_untranspose_rows": (C.c_int, [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, u8p, C.c_void_p]),
    "qvz_hip_qgen1": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, u8p, C.c_uint64, C.c_void_p]),
    "qvz_hip_kmeans_assign": (C.c_int, [u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, u8p, u8p, u32p,
                                        C.c_void_p]),
    "qvz_hip_kmeans_accumulate": (C.c_int, [u8p, u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, u64p,
                                            C.c_void_p]),
    "qvz_hip_encode_planes_release": (None, []),
    "qvz_hip_streams_overlap": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "qvz_hip_kmeans_step": (C.c_int, [u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, u8p, u8p, u32p, u64p, C.c_void_p]),
    "qvz_hip_kmeans_update": (C.c_int, [u64p, u32p, C.c_uint32, C.c_uint32, u8p, f64p, C.c_void_p, C.c_void_p]),
    "qvz_hip_kmeans_update64": (C.c_int, [u64p, u64p, C.c_uint32, C.c_uint32, u8p, f64p, C.c_void_p, C.c_void_p]),
    "qvz_hip_widen_u32": (C.c_int, [u32p, u64p, C.c_uint64, C.c_void_p]),
    "qvz_hip_generate_codebook64": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.c_int, C.c_double,
                                              C.c_double, C.c_void_p, C.c_void_p]),
    "qvz_hip_pmf_hist": (C.c_int, [u8p, u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, u32p, C.c_void_p]),
    "qvz_hip_well_words": (C.c_int, [C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, u32p, C.c_void_p]),
    "qvz_hip_well_draws": (C.c_int, [C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, C.c_uint32, u8p, C.c_uint64,
                                     C.c_void_p]),
    "qvz_hip_well_jump": (C.c_int, [C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32)]),
    "qvz_hip_quantize": (C.c_int, [u8p, u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, u16p, u16p, u8p,
                                   C.c_uint64, f64p, u8p, u8p, u8p, C.c_uint64, f64p, C.c_void_p, C.c_void_p]),
    "qvz_hip_pack_feed": (C.c_int, [u8p, u8p, C.c_uint64, C.c_uint32, C.c_uint64, u16p, C.c_void_p]),
    "qvz_hip_set_device": (C.c_int, [C.c_int]),
    "qvz_hip_mem_info": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "qvz_hip_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "qvz_hip_free": (C.c_int, [C.c_void_p]),
    "qvz_hip_malloc_host": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "qvz_hip_free_host": (C.c_int, [C.c_void_p]),
    "qvz_hip_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "qvz_hip_copy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "qvz_hip_copy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "qvz_hip_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "qvz_hip_stream_destroy": (C.c_int, [C.c_void_p]),
    "qvz_hip_stream_sync": (C.c_int, [C.c_void_p]),
    "qvz_hip_stream_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    # include/qvz_codebook.h
    "qvz_codebook_alloc": (C.c_void_p, [C.c_uint32]),
    "qvz_codebook_free": (None, [C.c_void_p]),
    "qvz_distortion_matrix": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "qvz_distortion_matrix_file": (C.c_int, [C.c_char_p, C.POINTER(C.c_double)]),
    "qvz_fastq_detect": (C.c_int, [C.c_void_p, C.c_uint64]),
    "qvz_fastq_qualities": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "qvz_hip_generate_codebook": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.c_int, C.c_double,
                                            C.c_double, C.c_void_p, C.c_void_p]),
    "qvz_hip_codebook_warmup": (C.c_int, []),
    "qvz_hip_codebook_exact_entropies": (C.c_int, []),
    "qvz_hip_codebook_exact_recheck_counts": (None, [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "qvz_hip_codebook_deferred_decisions": (C.c_uint64, []),
    "qvz_hip_marginals": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.c_void_p]),
    "qvz_codebook_tables": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "qvz_codebook_write_header": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "qvz_codebook_write": (C.c_size_t, [C.c_void_p, C.c_void_p]),
    # include/qvz_coder.h (host C)
    "qvz_coder_create": (C.c_void_p, [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "qvz_coder_destroy": (None, [C.c_void_p]),
    "qvz_coder_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "qvz_coder_finish": (None, [C.c_void_p]),
    "qvz_coder_bytes": (C.c_uint64, [C.c_void_p]),
    "qvz_coder_data": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "qvz_coder_drop": (None, [C.c_void_p, C.c_uint64]),
    "qvz_coder_encode_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]),
    "qvz_hip_model_session_open": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "qvz_hip_model_session_pitch": (C.c_uint32, [C.c_void_p]),
    "qvz_hip_model_session_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "qvz_hip_model_session_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "qvz_hip_model_session_close": (None, [C.c_void_p]),
    "qvz_shard_bounds": (None, [C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]),
    "qvz_encode_shards_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                           C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_int,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "qvz_hip_encode_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_uint64,
                                        C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64, C.c_void_p]),
    "qvz_hip_encode_planes_bounds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                               C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64,
                                               C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64, C.c_void_p]),
    "qvz_hip_decode_shards": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "qvz_hip_decode_tables_bytes": (C.c_uint64, [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "qvz_hip_decode_shards_bounds": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.c_uint32,
                                               C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64,
                                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "qvz_hip_encode_planes_prepare": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint64), C.c_uint64]),
    "qvz_hip_encode_planes_work_bytes": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
    "qvz_coder_segment_list": (C.c_uint32, [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]),
    "qvz_coder_segment_list2": (C.c_uint32, [C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]),
    "qvz_hip_encode_planes_last_ms": (None, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "qvz_hip_encode_planes_warmup": (C.c_int, []),
    "qvz_coder_rcp_one_step_ok_ex": (C.c_int, [C.POINTER(C.c_double)]),
    "qvz_coder_rcp_one_step_ok": (C.c_int, []),
    "qvz_model_build": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]),
    "qvz_shard_file_image": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64]),
    # include/qvz_encode.h (host C)
    "qvz_options_default": (None, [C.c_void_p]),
    "qvz_encode_buffer": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qvz_encode_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p]),
    "qvz_decode_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p]),
    "qvz_decode_shard_files": (C.c_int, [C.c_char_p, C.c_uint32, C.c_char_p, C.c_int, C.c_void_p]),
    "qvz_decode_set_slab_reads": (None, [C.c_uint64]),
    "qvz_decode_set_slab": (None, [C.c_uint64, C.c_uint32]),
    "qvz_codebook_read": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]),
    "qvz_encode_file_sharded": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "qvz_encode_last_error": (C.c_char_p, []),
    "qvz_encode_file_devices": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_void_p]),
    "qvz_encode_file_sweep": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]),
    "qvz_sweep_parse": (C.c_uint32, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "qvz_resident_open": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "qvz_resident_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32,
                                      C.c_int, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_void_p]),
    "qvz_resident_products": (C.c_int, [C.c_void_p, C.c_void_p]),
    "qvz_resident_pack": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "qvz_resident_close": (None, [C.c_void_p]),
    "qvz_hip_gather_reads": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "qvz_hip_sum_errors": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "qvz_hip_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "qvz_hip_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "qvz_hip_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "qvz_hip_event_destroy": (C.c_int, [C.c_void_p]),
    "qvz_egress_rows": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_char_p]),
    "qvz_ingest_rows": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32)]),
    "qvz_hip_scatter_streams": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]),
    "qvz_hip_gather_streams": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p, C.c_void_p]),
    "qvz_hip_well_jump_many": (C.c_int, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint32)]),
    "qvz_shard_limit": (C.c_uint32, [C.c_uint64]),
    # include/qvz_shardset.h (host C)
    "qvz_shardset_is": (C.c_int, [C.c_void_p, C.c_size_t]),
    "qvz_shardset_write_head": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]),
    "qvz_shardset_write": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]),
    "qvz_shardset_stat": (C.c_int, [C.c_char_p, C.c_void_p]),
    "qvz_shardset_split": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int]),
    "qvz_shardset_decode": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p]),
    "qvz_encode_file_shardset": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p]),
}

_lib = None


class QvzHipError(RuntimeError):
    pass


def lib():
    """The loaded C-ABI library. Raises (never falls back) when the HIP extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QvzHipError(f"{LIB_PATH} is missing: build it with `make` or __graft_entry__.build(); "
                              "qvz_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header/library drift
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise QvzHipError(f"qvz_hip error {rc}: {lib().qvz_hip_last_error().decode(errors='replace')}")
