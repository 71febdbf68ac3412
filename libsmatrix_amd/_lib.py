"""Loads the in-tree HIP library (lib/smatrix.so).  There is no CPU fallback:
a missing library is an ImportError that says how to build it."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "smatrix.so")
CSRC = os.path.join(HERE, "csrc")

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class Stats(C.Structure):
    # include/smatrix_batch.h smatrix_stats_t
    _fields_ = [(n, C.c_uint64) for n in (
        "rows", "dir_slots", "arena_units", "arena_mapped", "arena_free_units", "batches", "rounds",
        "deferred_ops", "rows_grown", "dir_grown", "rows_rebalanced", "long_probe_rounds", "scalar_cache_hits", "scalar_cache_flushes",
        "scalar_cache_flushed_cells", "bulk_rounds", "bulk_ops", "file_flushes", "file_rows_written", "file_leaked_bytes",
        "file_compactions", "spec_chains", "spec_refused", "file_bg_flushes", "cold_starts", "cold_keys", "clustered_mode", "set_located_by_fold", "flush_snapshots_refused")] + [
        ("kernel_ms", C.c_double * 4), ("kernel_launches", C.c_uint64 * 4), ("kernel_ops", C.c_uint64 * 4)] + [
        (n, C.c_double) for n in ("write_call_ms", "write_wait_ms", "write_alloc_ms", "last_write_call_ms", "last_write_wait_ms", "last_write_alloc_ms")]


class Handle(C.Structure):
    # include/smatrix.h smatrix_t
    _fields_ = [("fd", C.c_int), ("shutdown", C.c_int), ("fpos", C.c_uint64),
                ("mem", C.c_uint64), ("impl", C.c_void_p)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsmatrix_amd: %s is missing -- build it with `make -C %s` "
            "(or python -c 'import __graft_entry__ as g; g.build()').  "
            "This package has no CPU fallback." % (LIB_PATH, CSRC))
    lib = C.CDLL(LIB_PATH)
    H = C.POINTER(Handle)
    V = C.c_void_p
    sig = {
        # include/smatrix.h
        "smatrix_open": (H, [C.c_char_p]),
        "smatrix_close": (None, [H]),
        "smatrix_get": (C.c_uint32, [H, C.c_uint32, C.c_uint32]),
        "smatrix_set": (C.c_uint32, [H, C.c_uint32, C.c_uint32, C.c_uint32]),
        "smatrix_incr": (C.c_uint32, [H, C.c_uint32, C.c_uint32, C.c_uint32]),
        "smatrix_decr": (C.c_uint32, [H, C.c_uint32, C.c_uint32, C.c_uint32]),
        "smatrix_rowlen": (C.c_uint32, [H, C.c_uint32]),
        "smatrix_getrow": (C.c_uint32, [H, C.c_uint32, u32p, C.c_size_t]),
        # include/smatrix_batch.h
        "smatrix_apply_batch": (C.c_int, [H, C.c_int, C.c_size_t, u32p, u32p, u32p, u32p]),
        "smatrix_get_batch": (C.c_int, [H, C.c_size_t, u32p, u32p, u32p]),
        "smatrix_set_batch": (C.c_int, [H, C.c_size_t, u32p, u32p, u32p, u32p]),
        "smatrix_incr_batch": (C.c_int, [H, C.c_size_t, u32p, u32p, u32p, u32p]),
        "smatrix_decr_batch": (C.c_int, [H, C.c_size_t, u32p, u32p, u32p, u32p]),
        "smatrix_rowlen_batch": (C.c_int, [H, C.c_size_t, u32p, u32p]),
        "smatrix_getrow_batch": (C.c_int, [H, C.c_size_t, u32p, u64p, u32p, u32p]),
        "smatrix_apply_batch_dev": (C.c_int, [H, C.c_int, C.c_size_t, V, V, V, V, V]),
        "smatrix_apply_packed_dev": (C.c_int, [V, C.c_int, C.c_size_t, V, C.c_uint32, V, V]),
        "smatrix_rowlen_batch_dev": (C.c_int, [H, C.c_size_t, V, V, V]),
        "smatrix_getrow_batch_dev": (C.c_int, [H, C.c_size_t, V, V, V, V, V]),
        "smatrix_cf_neighbors_batch": (C.c_int, [H, C.c_size_t, u32p, u64p, u32p, C.POINTER(C.c_double), u32p]),
        "smatrix_cf_neighbors_batch_dev": (C.c_int, [H, C.c_size_t, V, V, V, V, V, V]),
        "smatrix_release_cached_memory": (None, []),
        "smatrix_reserve": (C.c_int, [H, C.c_uint64]),
        "smatrix_cf_topk_batch": (C.c_int, [H, C.c_size_t, u32p, C.c_uint32, u32p, C.POINTER(C.c_double), u32p]),
        "smatrix_cf_topk_batch_dev": (C.c_int, [H, C.c_size_t, V, C.c_uint32, V, V, V, V]),
        "smatrix_cf_import_sessions": (C.c_int, [H, C.c_size_t, u64p, u32p]),
        "smatrix_cf_import_sessions_dev": (C.c_int, [H, C.c_size_t, V, V, V, C.c_uint64, V]),
        "smatrix_stats": (None, [H, C.POINTER(Stats)]),
        "smatrix_stats_sz": (None, [H, C.POINTER(Stats), C.c_size_t]),
        "smatrix_profile": (None, [H, C.c_int]),
        "smatrix_flush": (C.c_int, [H]),
        "smatrix_compact": (C.c_int, [H]),
        "smatrix_row_info": (C.c_int, [H, C.c_uint32, u32p, u32p]),
        "smatrix_row_slots": (C.c_uint32, [H, C.c_uint32, u32p, C.c_uint32]),
        "smatrix_device_available": (C.c_int, []),
        # include/smatrix_shard.h
        "smatrix_shard_of": (C.c_uint32, [C.c_uint32, C.c_uint32]),
        "smatrix_partition_dev": (C.c_int, [C.c_size_t, V, V, V, C.c_uint32, u64p, V, V, V, V, V, V, C.c_uint32, V, V]),
        "smatrix_partition_packed_dev": (C.c_int, [C.c_size_t, V, V, V, C.c_uint32, u64p, V, V, V, V, C.c_uint32, V, V]),
        "smatrix_shard_mix": (C.c_uint32, [C.c_uint32]),
        "smatrix_place_slot": (C.c_uint32, [C.c_uint32, C.c_uint32]),
        "smatrix_displaced_rows": (C.c_size_t, [V, C.c_uint32, C.c_uint32, V, C.c_size_t]),
        "smatrix_unpack_dev": (C.c_int, [C.c_size_t, C.c_uint32, V, V, V, V, V]),
        "smatrix_gather_dev": (C.c_int, [C.c_size_t, V, V, V, V]),
        "smatrix_shard_unique_id": (C.c_int, [V]),
        "smatrix_shard_open": (V, [C.c_char_p, C.c_int, C.c_int, V]),
        "smatrix_shard_close": (None, [V]),
        "smatrix_shard_local": (H, [V]),
        "smatrix_shard_rank": (C.c_int, [V]),
        "smatrix_shard_nranks": (C.c_int, [V]),
        "smatrix_shard_ops_applied": (C.c_uint64, [V]),
        "smatrix_shard_set_placement": (C.c_int, [V, V, V, C.c_uint32]),
        "smatrix_shard_apply_dev": (C.c_int, [V, C.c_int, C.c_size_t, V, V, V, V, V]),
        "smatrix_shard_apply_then_get_dev": (C.c_int, [V, C.c_int, C.c_size_t, V, V, V, V, V, V]),
        "smatrix_shard_transport": (C.c_char_p, [V]),
        "smatrix_shard_plan_json": (C.c_int, [V, V, C.c_size_t, C.c_uint64, C.c_int, C.c_int, V, C.c_size_t]),
        "smatrix_shard_get_placement": (C.c_int, [V, V, V, V, C.c_uint32, V]),
        "smatrix_shard_route_dev": (V, [V, C.c_int, C.c_size_t, V, V, V, C.c_int, V]),
        "smatrix_shard_apply_routed": (C.c_int, [V, V, C.c_int, V]),
        "smatrix_shard_finish": (C.c_int, [V, V, V, V]),
        "smatrix_shard_wait": (C.c_int, [V, V, V]),
        "smatrix_shard_rowlen_dev": (C.c_int, [V, C.c_size_t, V, V, V]),
        "smatrix_shard_getrow_dev": (C.c_int, [V, C.c_size_t, V, V, V, V, V]),
        # include/smx_probe.h
        "smx_probe_random_dev": (C.c_int, [V, C.c_size_t, C.c_size_t, C.c_int, C.c_uint64, V, V]),
        # include/smx_stream.h
        "smx_stream_new": (V, [C.c_int, C.c_uint64, C.c_uint32, C.c_double, C.c_int]),
        "smx_stream_free": (None, [V]),
        "smx_stream_fill": (None, [V, C.c_uint64, C.c_size_t, u32p, u32p]),
        "smx_stream_fill_device": (C.c_int, [V, C.c_uint64, C.c_size_t, V, V, V]),
        "smx_splitmix64_at": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "smx_fmix32": (C.c_uint32, [C.c_uint32]),
        "smx_stream_cdf": (C.POINTER(C.c_double), [V, u32p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError = the ABI is incomplete: fail loudly
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib
