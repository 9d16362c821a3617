"""ctypes binding of libgraphsage_amd.so (the C ABI declared in include/graphsage_amd.h).

The product path has NO CPU fallback: if the shared library is missing or a kernel reports an
error, a GraphsageAmdError is raised.  torch is used only as the owner of device memory.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_C", "libgraphsage_amd.so")

ACT_IDENTITY = 0
ACT_RELU = 1
LAW_IID, LAW_REFERENCE, LAW_DISTINCT = 0, 1, 2      # GS_LAW_* (sampling law of the CSR sampler)
SAMPLER_LAWS = {"iid": LAW_IID, "reference": LAW_REFERENCE, "distinct": LAW_DISTINCT}
GS_PEER_HANDLE_BYTES = 64
GS_ABI_VERSION = 10     # must equal GS_ABI_VERSION of include/graphsage_amd.h (struct layouts below mirror that header)


class GraphsageAmdError(RuntimeError):
    pass


_P = c_void_p  # every device pointer crosses the boundary as void*

# name -> (argtypes) ; every function returns int except gs_last_error / gs_abi_version
_PROTOS = {
    "gs_device_info": [POINTER(c_int), POINTER(c_int), ctypes.c_char_p, c_int],
    "gs_sample_padded": [_P, c_int64, c_int32, _P, c_int64, _P, c_int32, _P, _P],
    "gs_sample_uniform_csr": [_P, _P, c_int64, c_int32, _P, c_int64, c_int32, c_uint64, c_uint64, _P, c_uint32,
                              c_int64, c_int32, c_int32, _P, _P],
    "gs_select_batch": [_P, c_int64, _P, c_int64, _P, _P],
    "gs_advance_counter": [_P, c_uint64, _P],
    "gs_gather_rows": [_P, c_int64, _P, c_int64, c_int32, _P, c_int64, _P],
    "gs_gather_mean_fwd": [_P, c_int64, _P, c_int64, c_int32, c_int32, _P, c_int64, _P, _P, c_int64, _P],
    "gs_mean_bwd": [_P, c_int64, c_int64, c_int32, c_int32, c_float, _P, c_int64, _P, c_int64, c_int, _P],
    "gs_sage_dense_fwd": [_P, c_int64, _P, c_int32, _P, c_int64, _P, c_int32, c_int64, _P, c_int64, _P, c_int64,
                          c_int32, c_int, c_int, _P, _P, c_int64, _P],
    "gs_dense_wgrad": [_P, c_int64, _P, c_int32, _P, c_int64, c_int32, c_int32, c_int64, c_int32, _P, c_int64, _P],
    "gs_dense_dgrad": [_P, c_int64, c_int32, c_int32, c_int64, _P, c_int64, c_int32, _P, c_int64, c_int, _P],
    "gs_act_bwd": [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int, _P, c_int64, _P],
    "gs_colsum_slabs": [_P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, _P],
    "gs_gemm_f32": [c_int, c_int, c_int64, c_int32, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int, _P, c_int64, _P],
    "gs_abi_struct_sizes": [_P, c_int32],
    "gs_dense_pool_max_fwd": [_P, c_int64, _P, c_int32, c_int64, c_int32, _P, c_int64, c_int32, _P, _P, c_int64, _P, c_int64, _P],
    "gs_segment_max_fwd": [_P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, _P, c_int64, _P],
    "gs_segment_max_bwd": [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, _P],
    "gs_l2norm_fwd": [_P, c_int64, c_int64, c_int32, _P, c_int64, _P, _P],
    "gs_l2norm_bwd": [_P, c_int64, _P, c_int64, _P, c_int64, c_int32, _P, c_int64, _P],
    "gs_class_loss": [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int, _P, _P, c_int64, _P, c_int64, _P],
    "gs_reduce_slabs": [_P, c_int32, c_int64, c_int32, c_int32, c_int64, c_float, _P, c_int64, _P, c_int64, c_int, _P],
    "gs_adam_step": [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_float, _P, c_int32, _P],
    "gs_comm_available": [],
    "gs_comm_count": [_P, POINTER(c_int32)],
    "gs_comm_unique_id": [_P, c_int32],
    "gs_comm_init_rank": [POINTER(c_void_p), c_int32, c_int32, _P, c_int32],
    "gs_comm_allreduce_sum_f32": [_P, _P, c_int64, _P],
    "gs_comm_destroy": [_P],
    "gs_peer_create": [c_int64, c_int32, c_int32, c_int32, c_int64, POINTER(c_void_p)],
    "gs_peer_export": [_P, _P, c_int32],
    "gs_peer_attach": [_P, c_int32, _P, c_int32],
    "gs_peer_attach_local": [_P, _P],
    "gs_peer_allreduce_sum_f32": [_P, _P, c_int64, _P],
    "gs_peer_status": [_P, POINTER(c_int64), POINTER(c_int32)],
    "gs_peer_destroy": [_P],
    "gs_sum_scaled": [_P, c_int64, c_float, _P, c_int, _P],
    "gs_sumsq_scaled": [_P, c_int64, c_float, _P, c_int, _P],
    "gs_stream_create": [POINTER(c_void_p)],
    "gs_spin_us": [c_float, _P],
    "gs_stream_destroy": [_P],
    "gs_stream_sync": [_P],
    "gs_capture_begin": [_P],
    "gs_capture_end": [_P, POINTER(c_void_p)],
    "gs_graph_launch": [_P, _P],
    "gs_graph_destroy": [_P],
    "gs_event_create": [POINTER(c_void_p)],
    "gs_event_record": [_P, _P],
    "gs_stream_wait_event": [_P, _P],
    "gs_event_elapsed_ms": [_P, _P, POINTER(c_float)],
    "gs_event_destroy": [_P],
    "gs_build_csr_host": [_P, _P, _P, c_int64, c_int64, c_int, _P, _P, c_int64, POINTER(c_int64)],
    "gs_dense_wgrad_grouped": [_P, c_int32, _P],
    "gs_dense_wgrad_grouped_cogather": [_P, c_int32, _P, c_int32, _P],
    "gs_sage_dense_dgrad": [_P, c_int64, c_int64, c_int32, c_int, _P, c_int64, _P, c_int64, c_int32, _P, c_int64, _P],
    "gs_flat_reduce_adam": [_P, c_int32, _P, _P, _P, _P, c_int64, c_float, c_int, c_float, c_float, c_float, c_float,
                            c_float, c_float, _P, c_int32, _P, c_int64, c_float, _P, c_int, _P],
    "gs_sage_dense_fwd_stream": [_P, c_int64, _P, _P, c_int64, c_int32, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int, _P, _P,
                                 c_int64, _P, c_int32, _P],
    "gs_sage_dense_fwd_stream2": [_P, c_int64, _P, c_int32, _P, c_int64, c_int32, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int, _P,
                                  _P, c_int64, _P, c_int32, _P],
    "gs_dense_wgrad_grouped_stream": [_P, c_int32, _P, c_int32, _P],
    "gs_dense_wgrad_grouped_tiled3": [_P, c_int32, _P, c_int32, _P],
    "gs_dense_wgrad_grouped_tiled3_sample": [_P, c_int32, _P, c_int32, _P, _P],
    "gs_sage_dense_fwd_tiled3": [_P, c_int64, _P, c_int32, _P, c_int64, c_int32, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int, _P,
                                 _P, c_int64, _P, c_int32, _P],
    "gs_split_rows_bytes": [c_int32, c_int32, _P],
    "gs_split_rows": [_P, c_int64, c_int32, c_int32, _P, _P],
    "gs_dense_fwd_rows_split": [_P, c_int64, _P, c_int32, c_int64, _P, _P, c_int32, c_int, _P, _P, c_int64, _P],
    "gs_split_rows_f16_bytes": [c_int32, c_int32, _P],
    "gs_split_rows_f16": [_P, c_int64, c_int32, c_int32, _P, _P],
    "gs_split_table_f16_bytes": [c_int64, c_int32, _P, _P],
    "gs_split_table_f16": [_P, c_int64, c_int64, c_int32, _P, _P, _P],
    "gs_dense_fwd_rows_split16": [_P, _P, _P, c_int32, c_int64, _P, _P, c_int32, c_int, _P, _P, c_int64, _P, c_int64, _P],
    "gs_dense_fwd_rows_split_ws_bytes": [_P],
    "gs_dense_fwd_rows_split_ws": [_P, c_int64, _P, c_int32, c_int64, _P, _P, c_int32, c_int, _P, _P, c_int64, _P, c_int64, _P],
    "gs_flat_reduce_adam_sample": [_P, c_int32, _P, _P, _P, _P, c_int64, c_float, c_int, c_float, c_float, c_float, c_float,
                                   c_float, c_float, _P, c_int32, _P, c_int64, c_float, _P, c_int, _P, _P, c_int32, _P],
    "gs_sage_tail_supported": [c_int32, c_int32, c_int32],
    "gs_sage_tail_fwd_bwd": [_P, _P, c_int32, _P],
    "gs_sage_tail_z": [_P, _P, c_int32, _P],
    "gs_sage_tail_dh0": [_P, _P, c_int32, _P],
    "gs_dropout_rows": [_P, c_int64, _P, c_int64, c_int32, _P, _P, c_int64, _P],
    "gs_gather_mean_dropout_fwd": [_P, c_int64, _P, c_int64, c_int32, c_int32, _P, c_int64, _P, _P, c_int64, _P, _P],
    "gs_scatter_add_rows": [_P, c_int64, c_int64, c_int32, c_int32, c_float, _P, _P, c_int64, _P],
    "gs_copy_cols": [_P, c_int64, _P, c_int64, c_int64, c_int32, _P],
    "gs_input_grad_pull": [_P, _P],
    "gs_advance_counters": [_P, c_uint64, _P, c_uint64, _P, c_uint64, _P],
    "gs_head_fwd_bwd": [_P, c_int64, c_int64, c_int32, _P, c_int64, _P, _P, c_int64, c_int32, c_int, _P, c_int64, _P,
                        c_int64, _P, c_int64, _P, c_int64, _P, _P, c_int64, _P],
    "gs_sample_fanout_csr": [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, c_int64, c_uint64, c_uint64, _P, c_uint32,
                             c_int64, _P, c_int64, _P, _P, c_int64, c_int32, _P, c_int64, c_int32, c_int32, _P],
    "gs_finalize_step": [_P, c_int64, c_float, _P, c_int, _P, c_uint64, _P, c_uint64, _P, c_uint64, _P],
    "gs_sage_dense_fwd_cogather": [_P, c_int64, _P, c_int32, _P, c_int64, _P, c_int32, c_int64, _P, c_int64, _P, c_int64,
                                   c_int32, c_int, c_int, _P, _P, c_int64, _P, c_int32, _P],
    "gs_unsup_stage": [_P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_uint64, _P, c_int64, _P, _P],
    "gs_linkpred_fwd_bwd": [_P, c_int64, c_int64, c_int32, c_int32, c_float, c_float, _P, _P, _P, c_int64, _P, c_int64,
                            _P, POINTER(c_int32), _P],
    "gs_linkpred_norm_fwd_bwd": [_P, c_int64, c_int64, c_int32, c_int32, c_float, c_float, _P, c_int64, _P, _P, _P, c_int64,
                                 _P, c_int64, _P, _P],
    "gs_linkpred_norm_fwd_bwd_step": [_P, c_int64, c_int64, c_int32, c_int32, c_float, c_float, _P, c_int64, _P, _P, _P, c_int64,
                                      _P, c_int64, _P, _P, c_int, _P, _P, c_uint64, _P, c_uint64, _P, c_uint64, _P],
    "gs_linkpred_tail_supported": [c_int32, c_int32, c_int32],
    "gs_linkpred_tail": [_P, _P, c_int32, _P],
    "gs_linkpred_tail_neg": [_P, _P, c_int, _P, _P, c_uint64, _P, c_uint64, _P, c_uint64, _P, c_int32, _P],
    "gs_unique_ids": [_P, c_int64, c_int64, _P, _P, _P, _P, _P, _P],
    "gs_dense_fwd_rows_dev": [_P, c_int64, _P, c_int32, c_int64, _P, _P, c_int64, c_int32, c_int, _P, _P, c_int64, _P],
    "gs_segment_max_gather_fwd": [_P, c_int64, _P, c_int64, c_int32, c_int32, _P, c_int64, _P, c_int64, _P],
    "gs_sample_fanout_desc": [_P, _P],
    "gs_build_padded_table": [_P, _P, c_int64, c_int32, c_int32, c_uint64, _P, _P],
    "gs_finalize_step2": [_P, c_int64, c_float, _P, c_int, _P, c_float, _P, _P, c_uint64, _P, c_uint64, _P, c_uint64, _P],
    "gs_maxpool_sparse_wgrad": [_P, c_int64, _P, c_int64, c_int32, c_int32, _P, c_int64, _P, c_int64, c_int32, c_int32, _P,
                                c_int64, _P],
    "gs_stage_batch": [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int32, _P, c_int64, _P],
}


class WgradDesc(ctypes.Structure):
    """struct gs_wgrad_desc (include/graphsage_amd.h)"""
    _fields_ = [("A", c_void_p), ("a_idx", c_void_p), ("dZ", c_void_p), ("slabs", c_void_p),
                ("lda", c_int64), ("ldz", c_int64), ("ld_slab", c_int64), ("n", c_int64),
                ("d", c_int32), ("col0", c_int32), ("out_dim", c_int32), ("n_slabs", c_int32), ("a_rows", c_int64)]


class GatherDesc(ctypes.Structure):
    """struct gs_gather_desc (include/graphsage_amd.h)"""
    _fields_ = [("X", c_void_p), ("idx", c_void_p), ("self_src", c_void_p), ("self_idx", c_void_p), ("out", c_void_p),
                ("ldx", c_int64), ("ld_self", c_int64), ("ldo", c_int64), ("n", c_int64), ("s", c_int32), ("d", c_int32)]


class Dropout(ctypes.Structure):
    """struct gs_dropout (include/graphsage_amd.h)"""
    _fields_ = [("seed", c_uint64), ("clock_dev", c_void_p), ("site", c_uint32), ("rate", c_float), ("row0", c_int64),
                ("keep_bits", c_void_p), ("keep_ld", c_int64)]


GS_PULL_MAX = 6


class PullDesc(ctypes.Structure):
    """struct gs_pull_desc (include/graphsage_amd.h)"""
    _fields_ = [("d_self", c_void_p), ("ld_self", c_int64), ("n_self", c_int64), ("n_seg", c_int32), ("d", c_int32),
                ("src", c_void_p * GS_PULL_MAX), ("ld_src", c_int64 * GS_PULL_MAX), ("row0", c_int64 * GS_PULL_MAX),
                ("n", c_int64 * GS_PULL_MAX), ("s", c_int32 * GS_PULL_MAX), ("scale", c_float * GS_PULL_MAX),
                ("mask_y", c_void_p), ("ldy", c_int64), ("out", c_void_p), ("ldo", c_int64), ("rows", c_int64)]


class TailDesc(ctypes.Structure):
    """struct gs_tail_desc (include/graphsage_amd.h)"""
    _fields_ = [("h0", c_void_p), ("ldh", c_int64), ("n", c_int64),
                ("W_self", c_void_p), ("ldws", c_int64), ("W_neigh", c_void_p), ("ldwn", c_int64),
                ("W_head", c_void_p), ("ldwh", c_int64), ("b_head", c_void_p),
                ("labels", c_void_p), ("ldlab", c_int64),
                ("means", c_void_p), ("ldm", c_int64), ("z", c_void_p), ("ldz", c_int64), ("y", c_void_p), ("ldy", c_int64),
                ("logits", c_void_p), ("ldlo", c_int64), ("preds", c_void_p), ("ldp", c_int64),
                ("dlogits", c_void_p), ("lddl", c_int64), ("loss_rows", c_void_p),
                ("dz", c_void_p), ("lddz", c_int64), ("d_h0", c_void_p), ("lddh", c_int64),
                ("c0", c_void_p), ("d0", c_uint64), ("c1", c_void_p), ("d1", c_uint64), ("c2", c_void_p), ("d2", c_uint64),
                ("s", c_int32), ("d_in", c_int32), ("out_dim", c_int32), ("C", c_int32), ("sigmoid", c_int32),
                ("train", c_int32), ("sync", c_void_p), ("z_ready", c_int32), ("gcn", c_int32),
                ("ids_copy_src", c_void_p), ("ids_copy_dst", c_void_p), ("ids_copy_n", c_int64)]


class LpTailDesc(ctypes.Structure):
    """struct gs_lp_tail_desc (include/graphsage_amd.h)"""
    _fields_ = [("h0", c_void_p), ("ldh", c_int64), ("B", c_int64), ("n_neg", c_int32), ("s", c_int32), ("d_in", c_int32),
                ("out_dim", c_int32), ("train", c_int32),
                ("W_self", c_void_p), ("ldws", c_int64), ("W_neigh", c_void_p), ("ldwn", c_int64),
                ("means", c_void_p), ("ldm", c_int64), ("z", c_void_p), ("ldz", c_int64), ("y", c_void_p), ("ldy", c_int64),
                ("dz", c_void_p), ("lddz", c_int64), ("d_h0", c_void_p), ("lddh", c_int64),
                ("loss_rows", c_void_p), ("rr_rows", c_void_p), ("aff_all", c_void_p), ("ld_aff", c_int64),
                ("neg_slabs", c_void_p), ("neg_weight", c_float), ("scale", c_float), ("sync", c_void_p)]


class FanoutDesc(ctypes.Structure):
    """struct gs_fanout_desc (include/graphsage_amd.h)"""
    _fields_ = [("rowptr", c_void_p), ("col", c_void_p), ("n_nodes", c_int64),
                ("ids_all", c_void_p), ("B", c_int64),
                ("seed", c_uint64), ("step", c_uint64), ("step_dev", c_void_p),
                ("root_offset", c_int64),
                ("order", c_void_p), ("n_order", c_int64), ("cursor_dev", c_void_p),
                ("label_table", c_void_p), ("ld_table", c_int64),
                ("labels_out", c_void_p), ("ld_out", c_int64),
                ("offsets", c_int64 * 4), ("fan", c_int32 * 3),
                ("pad_id", c_int32), ("n_hops", c_int32), ("C", c_int32), ("hop0", c_uint32),
                ("law", c_int32), ("max_degree", c_int32),
                ("pairs", c_void_p), ("n_pairs", c_int64), ("n_pair_roots", c_int64),
                ("cdf", c_void_p), ("guide", c_void_p), ("n_cdf", c_int64),
                ("n_neg", c_int32), ("guide_bits", c_int32), ("neg_seed", c_uint64), ("padded_table", c_void_p),
                ("seg_begin", c_int64 * 2)]


class VarDesc(ctypes.Structure):
    """struct gs_var_desc (include/graphsage_amd.h)"""
    _fields_ = [("offset", c_int64), ("size", c_int64), ("slabs", c_void_p), ("n_slabs", c_int32), ("decay", c_int32),
                ("clear", c_int32), ("reserved_", c_int32)]

EXPORTED_SYMBOLS = sorted(list(_PROTOS.keys()) + ["gs_last_error", "gs_abi_version"])

_lib = None


def load(build_if_missing=True):
    """Load (building first if hipcc is available and the sources changed) and return the CDLL."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64 (SONAME libamdhip64.so.7); it must be mapped BEFORE this library so that the
    # dynamic linker binds our NEEDED libamdhip64.so.7 to that same runtime.  Loading ours first would map
    # /opt/rocm's copy and torch would then bring in a second HIP runtime (its NEEDED entry is the un-versioned
    # name) -- two runtimes in one process do not share streams or allocations ("no ROCm-capable device").
    import torch  # noqa: F401
    alt = os.environ.get("GS_LIB")          # A/B hook: load this prebuilt library instead (no build / digest check)
    if alt:
        build_if_missing = False
    if build_if_missing:
        try:
            from . import build as _build
            _build.build()
        except Exception as e:  # no hipcc on this box: fall through to the prebuilt library check
            if not os.path.exists(LIB_PATH):
                raise GraphsageAmdError("libgraphsage_amd.so is missing and could not be built: %s" % e)
    path = alt or LIB_PATH
    if not os.path.exists(path):
        raise GraphsageAmdError("HIP extension not found at %s (run python -m graphsage_amd.build)" % path)
    lib = ctypes.CDLL(path)
    lib.gs_last_error.restype = c_char_p
    lib.gs_last_error.argtypes = []
    lib.gs_abi_version.restype = c_int
    lib.gs_abi_version.argtypes = []
    got = lib.gs_abi_version()
    if got != GS_ABI_VERSION:
        raise GraphsageAmdError("%s has ABI version %d but this package binds version %d: the ctypes struct layouts "
                                "would not match (rebuild with python -m graphsage_amd.build --force)"
                                % (LIB_PATH, got, GS_ABI_VERSION))
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = c_int
        fn.argtypes = argtypes
    # struct layouts: the library's sizeof() of every descriptor must equal the ctypes mirror's
    mirrors = [GatherDesc, WgradDesc, VarDesc, FanoutDesc, TailDesc, Dropout, PullDesc, LpTailDesc]
    sizes = (c_int32 * 16)()
    n = lib.gs_abi_struct_sizes(sizes, 16)
    if n != len(mirrors):
        raise GraphsageAmdError("%s exports %d descriptor structs, this package mirrors %d" % (LIB_PATH, n, len(mirrors)))
    for cls, size in zip(mirrors, sizes):
        if ctypes.sizeof(cls) != size:
            raise GraphsageAmdError("struct layout mismatch: sizeof(%s) is %d in %s but %d in graphsage_amd/_lib.py"
                                    % (cls.__doc__.split()[1], size, LIB_PATH, ctypes.sizeof(cls)))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gs_last_error()
        raise GraphsageAmdError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


_DEBUG_SYNC = os.environ.get("GS_DEBUG_SYNC", "0") == "1"


def call(name, *args):
    lib = load()
    if _DEBUG_SYNC:
        import sys
        import torch
        sys.stderr.write("[gs] %s %s\n" % (name, " ".join(str(a) for a in args)))
        sys.stderr.flush()
    rc = getattr(lib, name)(*args)
    check(rc, name)
    if _DEBUG_SYNC and not name.startswith(("gs_capture", "gs_graph", "gs_stream", "gs_event")):
        torch.cuda.synchronize()


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  Refuses CPU tensors: no CPU fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GraphsageAmdError("graphsage_amd kernels need device (HIP) tensors; got a CPU tensor")
    return t.data_ptr()


def host_ptr(a):
    """Pointer of a contiguous numpy array (host-side entry points only)."""
    if a is None:
        return None
    return a.ctypes.data
