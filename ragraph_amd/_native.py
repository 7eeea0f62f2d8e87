"""ctypes binding of libragraph_hip.so (include/ragraph_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared object is missing, or the
process has no gfx950 device, every compute call raises `RagraphNativeError`.

`import torch` happens before the dlopen so that the HIP runtime the library resolves (SONAME libamdhip64.so.7) is the
one PyTorch already loaded: device pointers and streams are then shared between torch (memory, streams,
torch.distributed) and these kernels.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede the dlopen, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("RAGRAPH_HIP_SO", os.path.join(CSRC, "libragraph_hip.so"))  # override: A/B builds

OK, EINVAL, EUNSUPPORTED, EWORKSPACE, EDEVICE = 0, -1, -2, -3, -4
ACT_NONE, ACT_RELU, ACT_PRELU, ACT_LEAKY, ACT_ELU = 0, 1, 2, 3, 4
TOPK_MAX = 64

_vp, _i64, _i32, _f32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); exactly the entry points include/ragraph_hip.h declares.
SIGNATURES = {
    "ragraph_abi_version": (_i32, []),
    "ragraph_last_error": (ctypes.c_char_p, []),
    "ragraph_device_check": (_i32, []),
    "ragraph_normalize_rows_f32": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "ragraph_topk_cosine_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "ragraph_topk_cosine_f32": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_pack_keys_f32": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "ragraph_keys_bf16_rows": (_i64, [_i64]),
    "ragraph_keys_to_bf16": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "ragraph_filter_profile_create": (_vp, []),
    "ragraph_filter_profile_destroy": (None, [_vp]),
    "ragraph_filter_profile_attach": (_vp, [_vp]),
    "ragraph_filter_profile_last_ms": (ctypes.c_float, [_vp]),
    "ragraph_topk_cosine_filtered_set_prior": (ctypes.c_float, [_f32]),
    "ragraph_filter_profile_levels": (_i32, [_vp, _vp, _vp, _vp]),
    "ragraph_topk_cosine_filtered_cap": (_i32, [_i32]),
    "ragraph_topk_cosine_filtered_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "ragraph_topk_cosine_filtered_stats_offset": (_sz, [_sz]),
    "ragraph_topk_cosine_filtered_sharded_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32, _i32]),
    "ragraph_topk_cosine_filtered_plan": (_i32, [_i64, _i64, _i32, _i32, _vp]),
    "ragraph_topk_cosine_filtered_i8_levels": (_i32, [_i64, _i64, _i32, _i32]),
    "ragraph_topk_cosine_filtered_sharded_speculates": (_i32, [_i64, _i64, _i32, _i32, _i32]),
    "ragraph_verify_merged_prior_f32": (_i32, [_vp, _i64, _i32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "ragraph_topk_cosine_filtered_max_i8_levels": (_i32, [_i32]),
    "ragraph_topk_cosine_filtered_f32": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp,
                                                _sz, _vp]),
    "ragraph_topk_cosine_filtered_sharded_f32": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp,
                                                        _vp, _sz, _vp, _i64, _vp, _vp, _vp, _i32]),
    "ragraph_topk_cosine_small_ok": (_i32, [_i64, _i64, _i32, _i32]),
    "ragraph_topk_cosine_small_state_bytes": (_sz, []),
    "ragraph_topk_cosine_small_prefix_keys": (_i64, [_i64, _i64, _i32, _vp]),
    "ragraph_topk_cosine_small_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "ragraph_topk_cosine_small_f32": (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_topk_cosine_fused_ok": (_i32, [_i64, _i64, _i32, _i32]),
    "ragraph_topk_cosine_fused_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "ragraph_topk_cosine_fused_f32": (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_theta_sharpen_f32": (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    "ragraph_topk_cosine_bank_f32": (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_topk_merge_f32": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    "ragraph_dedup_rows_workspace_bytes": (_sz, [_i64]),
    "ragraph_dedup_rows_f32": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_topk_expand_groups_f32": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp, _i64, _i64, _i32, _i64, _vp, _vp, _vp]),
    "ragraph_gather_rows_f32": (_i32, [_vp, _i64, _i32, _vp, _i64, _i64, _vp, _vp]),
    "ragraph_gather_reduce_f32": (_i32, [_vp, _i32, _vp, _i32, _i64, _vp, _i64, _i32, _i64, _f32, _vp, _vp, _vp]),
    "ragraph_gather_reduce_mix_f32": (_i32, [_vp, _i32, _vp, _i32, _i64, _vp, _i64, _i32, _i64, _f32, _vp, _f32, _f32, _vp, _vp, _vp]),
    "ragraph_linear_f32": (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _i32, _f32, _vp, _vp]),
    "ragraph_linear_tn_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "ragraph_linear_tn_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _sz, _vp]),
    "ragraph_spmm_csr_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _i32, _f32, _f32, _vp, _vp, _vp]),
    "ragraph_radix_sort_workspace_bytes": (_sz, [_i64, _i32]),
    "ragraph_radix_sort_u64": (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _sz, _vp]),
    "ragraph_scan_workspace_bytes": (_sz, [_i64]),
    "ragraph_scan_sum_i32": (_i32, [_vp, _vp, _i64, _i32, _vp, _sz, _vp]),
    "ragraph_coo_to_csr_workspace_bytes": (_sz, [_i64, _i64]),
    "ragraph_coo_to_csr_i64": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_csr_row_ids_i64": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "ragraph_mask_positions_workspace_bytes": (_sz, [_i64]),
    "ragraph_mask_positions_i64": (_i32, [_vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_spmm_csr_panels_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _f32, _vp, _i32, _vp]),
    "ragraph_spmm_linear_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _i32, _f32, _vp, _vp]),
    "ragraph_spmm_csr_tiled_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _vp, _i64, _i32, _i32, _i32, _f32, _vp, _i32, _vp]),
    "ragraph_csr_row_normalize_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "ragraph_segment_softmax_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "ragraph_segment_softmax_ws_f32": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "ragraph_sparse_workspace_bytes": (_sz, [_i64, _i32]),
    "ragraph_spmm_csr_ws_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _i32, _f32, _f32, _vp, _vp, _i64, _vp, _sz,
                                       _vp]),
    "ragraph_axpby_f32": (_i32, [_vp, _f32, _vp, _f32, _i64, _vp, _vp]),
    "ragraph_softmax_mix_f32": (_i32, [_vp, _vp, _i64, _i32, _f32, _i32, _vp, _vp]),
    "ragraph_fuse_decode_f32": (_i32, [_vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _i32, _f32, _vp, _vp, _i32, _vp, _f32,
                                       _vp, _vp]),
    "ragraph_segment_reduce_f32": (_i32, [_vp, _i32, _vp, _i64, _vp, _i32, _vp, _vp]),
    "ragraph_proto_cosine_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _vp]),
    "ragraph_sigmoid_gate_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "ragraph_time_rescale_f32": (_i32, [_vp, _i64, _f32, _f32, _vp, _vp]),
    "ragraph_topk_rows_f32": (_i32, [_vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp]),
    "ragraph_topk_select_rows_f32": (_i32, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "ragraph_topk_select_rows_workspace_bytes": (_sz, [_i64, _i64]),
    "ragraph_topk_select_rows_ws_f32": (_i32, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_scatter_fill_f32": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _f32, _vp]),
    "ragraph_floyd_warshall_f32": (_i32, [_vp, _i32, _vp, _vp]),
    "ragraph_position_code_f32": (_i32, [_vp, _i32, _vp, _i32, _f32, _vp, _vp]),
    "ragraph_position_codes_csr_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _f32, _vp, _vp, _vp]),
    "ragraph_act_grad_f32": (_i32, [_vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp]),
    "ragraph_sigmoid_gate_grad_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ragraph_softmax_grad_f32": (_i32, [_vp, _vp, _i64, _i32, _f32, _vp, _vp]),
    "ragraph_mul_cols_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "ragraph_mul_cols_act_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp]),
    "ragraph_mul_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "ragraph_proto_cosine_grad_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ragraph_proto_cosine_grad_proto_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "ragraph_proto_cosine_grad_proto_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_axpby_dev_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i64, _vp, _vp]),
    "ragraph_ingest_workspace_bytes": (_sz, [_i64, _i64]),
    "ragraph_csr_sym_normalized_f32": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_binorm_edges_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_pagerank_workspace_bytes": (_sz, [_i64, _i64]),
    "ragraph_pagerank_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "ragraph_sample_prob_f32": (_i32, [_vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp]),
    "ragraph_csr_row_sums_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "ragraph_position_codes_batch_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _f32, _vp, _vp, _vp]),
}


class RagraphNativeError(RuntimeError):
    """The HIP library is missing, has no usable device, or a call returned an error code."""


_lib = None


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libragraph_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return SO_PATH


def lib() -> ctypes.CDLL:
    """dlopen the library and type every symbol of the header.  Does not touch the GPU."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RagraphNativeError(
                f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(ragraph_amd has no CPU fallback)")
        cdll = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if cdll.ragraph_abi_version() != 1:
            raise RagraphNativeError("libragraph_hip.so ABI version mismatch")
        _lib = cdll
    return _lib


def last_error() -> str:
    return lib().ragraph_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != OK:
        raise RagraphNativeError(f"{what} failed (code {rc}): {last_error()}")


def require_device() -> None:
    """Raise unless this process can run the kernels (gfx950 visible).  Called by every tensor wrapper."""
    if not torch.cuda.is_available():
        raise RagraphNativeError("no ROCm device visible to PyTorch; ragraph_amd has no CPU fallback")
    check(lib().ragraph_device_check(), "ragraph_device_check")
