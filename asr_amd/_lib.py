"""ctypes binding of libds2hip.so (C-ABI declared in include/ds2hip.h).

There is NO fallback: if the shared library is missing or a kernel call fails this raises.  The
library is built in-tree (asr_amd/lib/libds2hip.so) by `__graft_entry__.build()` / `make -C
asr_amd/csrc`, so it travels with the source snapshot to the GPU box.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DS2_LIB_PATH") or os.path.join(_HERE, "lib", "libds2hip.so")   # (override: same-box A/B of two builds)

_lib: Optional[C.CDLL] = None

vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes)   — mirrors include/ds2hip.h one to one
SIGNATURES = {
    "ds2_version": (C.c_char_p, []),
    "ds2_last_error": (C.c_char_p, []),
    "ds2_device_info": (i32, [C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32]),
    "ds2_debug_flags": (i32, [vp, i32]),
    "ds2_rnn_ctx_init": (i32, [vp, vp, vp, vp]),
    "ds2_memset_async": (i32, [vp, i32, sz, vp]),
    "ds2_ablation_build": (i32, []),
    "ds2_rnn_persistent_status": (i32, [vp, vp]),
    "ds2_rnn_persistent_counters": (i32, [vp, vp]),
    "ds2_rnn_poison_if_starved": (i32, [vp, vp, sz, vp]),
    "ds2_rnn_poison_seen": (i32, [vp]),
    "ds2_rnn_step_gate": (i32, [vp, vp, vp, vp]),
    "ds2_rnn_persistent_enable": (i32, [vp, i32, i32]),
    "ds2_rnn_last_path": (i32, [vp]),
    "ds2_rnn_bwd_ksplit_footprint": (i32, [i32, i32, C.POINTER(i32)]),
    "ds2_gemm_f32_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "ds2_gemm_f32": (i32, [i32, i32, i32, i32, i32, vp, i32, i64, vp, i32, i64, vp, i32, i64, vp, i32, i32, i32, vp, sz, vp]),
    "ds2_gemm_bf16_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "ds2_conv1_fwd_bf16_stat_blocks": (i32, [i32, i32, i32]),
    "ds2_conv1_fwd_bf16_stats": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "ds2_conv2_fwd_bf16_stat_blocks": (i32, [i32, i32, i32]),
    "ds2_conv2_fwd_bf16_stats": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "ds2_chanstats_from_partials_workspace_bytes": (sz, []),
    "ds2_chanstats_from_partials": (i32, [vp, i32, i32, f64, vp, vp, vp, vp, f32, vp, sz, vp]),
    "ds2_gemm_bf16_tn": (i32, [i32, i32, i32, vp, i32, i64, vp, i32, i64, vp, i32, i64, i32, i32, i32, vp, sz, vp]),
    "ds2_gemm_bf16_tn_group": (i32, [i32, vp, i32, vp]),
    "ds2_gemm_bf16_tn_splitk_group_workspace_bytes": (sz, [i32, vp, i32]),
    "ds2_gemm_bf16_tn_splitk_group": (i32, [i32, vp, i32, vp, sz, vp]),
    "ds2_gemm_bf16_tn_splitk_group_ep": (i32, [i32, vp, i32, i32, vp, vp, vp, vp, sz, vp]),
    "ds2_gemm_bf16_nt": (i32, [i32, i32, i32, vp, i32, i64, vp, i32, i64, vp, i32, i64, vp, i32, i32, i32, vp, sz, vp]),
    "ds2_cast_bf16": (i32, [vp, i32, vp, i32, i32, i32, vp]),
    "ds2_gemm_bf16_nt_obf16": (i32, [i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp]),
    "ds2_cast_f32_from_bf16": (i32, [vp, vp, i64, vp]),
    "ds2_rnn_fwd_x": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    "ds2_split_bf16": (i32, [vp, i32, vp, i32, i32, i32, i32, vp]),
    "ds2_cast_transpose_bf16": (i32, [vp, i32, vp, i32, i32, i32, vp]),
    "ds2_cast_bf16_both_workspace_bytes": (sz, [i32, i32]),
    "ds2_cast_bf16_both": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, vp, vp, sz, vp]),
    "ds2_transpose_bf16": (i32, [vp, i32, vp, i32, i32, i32, vp, vp, sz, vp]),
    "ds2_colreduce_workspace_bytes": (sz, [i32, i32]),
    "ds2_colstats_f32": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, f32, vp, sz, vp]),
    "ds2_add_colstats_f32": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, f32, vp, sz, vp]),
    "ds2_colsum_f32": (i32, [vp, i32, i32, i32, vp, vp, vp, sz, vp]),
    "ds2_bn1d_apply_f32": (i32, [vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, f32, vp]),
    "ds2_bn1d_apply_bf16": (i32, [vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, f32, vp]),
    "ds2_bn1d_bwd_f32": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, sz, vp]),
    "ds2_bn1d_bwd_xbf16": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, sz, vp]),
    "ds2_center_colstats": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, f32, vp, sz, vp]),
    "ds2_wih_fold_bf16": (i32, [vp, i32, vp, i32, i32, vp, vp, vp, vp, f32, vp, i32, vp, vp, vp, vp]),
    "ds2_scale_rank1_f32": (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
    "ds2_chanreduce_workspace_bytes": (sz, [i32]),
    "ds2_bn2d_stats_f32": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, f32, vp, sz, vp]),
    "ds2_bn2d_act_fwd_f32": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, vp]),
    "ds2_bn2d_act_bwd_f32": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, sz, vp]),
    "ds2_bn2d_act_fwd_fused": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp]),
    "ds2_bn2d_act_collapse": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, i32, vp]),
    "ds2_bn2d_act_bwd_fused_workspace_bytes": (sz, [i32, i32, i32]),
    "ds2_bn2d_act_bwd_fused": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "ds2_transpose_bft_f32": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ds2_transpose2d_f32": (i32, [vp, i32, i64, vp, i32, i64, i32, i32, i32, vp]),
    "ds2_conv_dims": (None, [i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "ds2_conv_packed_floats": (sz, [i32]),
    "ds2_conv_pack_f32": (i32, [vp, vp, vp, vp, vp, vp]),
    "ds2_conv1_fwd_f32": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ds2_conv2_fwd_f32": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ds2_conv2_dgrad_f32": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "ds2_conv_wgrad_workspace_bytes": (sz, [i32, i32, i32]),
    "ds2_conv1_wgrad_f32": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "ds2_conv2_wgrad_f32": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "ds2_conv2_bf16_packed_bytes": (sz, [i32]),
    "ds2_conv2_pack_bf16": (i32, [vp, vp, vp, vp, vp]),
    "ds2_nhwc_bf16_f32": (i32, [vp, vp, i32, i32, i32, vp]),
    "ds2_conv2_fwd_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ds2_conv2_dgrad_bf16": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "ds2_conv_padded_pitch": (i32, [i32]),
    "ds2_padcast_bf16": (i32, [vp, vp, i64, i32, vp]),
    "ds2_conv2_wgrad_bf16_workspace_bytes": (sz, [i32, i32]),
    "ds2_conv2_wgrad_bf16": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "ds2_conv2_wgrad_nhwc_bf16": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "ds2_rnn_packed_bytes": (sz, [i32, i32, i32, i32]),
    "ds2_rnn_pack_whh": (i32, [i32, vp, vp, vp, i32, i32, vp]),
    "ds2_rnn_fwd_workspace_bytes": (sz, [i32, i32, i32]),
    "ds2_rnn_fwd_ex": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "ds2_rnn_fwd": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "ds2_rnn_bwd_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "ds2_rnn_bwd_ex": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "ds2_rnn_bwd_bn": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "ds2_rnn_bwd_bn_xbf16": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "ds2_rnn_bias_grads": (i32, [i32, vp, i32, i32, vp, vp, vp]),
    "ds2_rnn_bwd": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "ds2_ctc_workspace_bytes": (sz, [i32, i32, i32]),
    "ds2_ctc_loss_f32": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, i32, f32, vp, sz, vp]),
    "ds2_ctc_batch_mean_f32": (i32, [vp, i32, vp, vp]),
    "ds2_add_i64": (i32, [vp, i32, i64, vp]),
    "ds2_softmax_rows_f32": (i32, [vp, i32, vp, i32, i32, i32, vp]),
    "ds2_greedy_decode_workspace_bytes": (sz, [i32, i32]),
    "ds2_greedy_decode_f32": (i32, [vp, i64, i64, i32, i32, i32, vp, i32, vp, vp, vp, vp, sz, vp]),
    "ds2_conv1_bf16_row_pitch": (i32, [i32]),
    "ds2_conv1_bf16_bytes": (sz, [i32, i32, i32, i32]),
    "ds2_conv1_pack_bf16": (i32, [vp, vp, vp]),
    "ds2_conv1_gather_bf16": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "ds2_conv1_fwd_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "ds2_conv1_wgrad_bf16_workspace_bytes": (sz, [i32, i32]),
    "ds2_conv1_wgrad_bf16": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "ds2_spectrogram_frames": (i32, [i32, i32]),
    "ds2_spectrogram_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "ds2_spectrogram_f32": (i32, [vp, i64, vp, i32, i32, i32, i32, vp, i32, i32, vp, vp, sz, vp]),
    "ds2_adamw_f32": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
    "ds2_adamw_gated_f32": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp, vp]),
    "ds2_scale_f32": (i32, [vp, i64, f32, vp]),
    "ds2_bf16_residual_f32": (i32, [vp, vp, i64, vp]),
    "ds2_sum3_f32": (i32, [vp, vp, vp, vp, i64, vp]),
}


class TnProblem(C.Structure):
    """`ds2_tn_problem` of include/ds2hip.h (one product of ds2_gemm_bf16_tn_group)."""
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("M", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32), ("ldc", i32)]


class DS2LibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libds2hip.so (once).  Raises DS2LibraryError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DS2LibraryError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'  or  make -C asr_amd/csrc). "
            "asr_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => header/library mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ds2_last_error().decode(errors="replace")
        raise DS2LibraryError(f"libds2hip call failed{(' in ' + what) if what else ''}: {msg}")


def version() -> str:
    return load().ds2_version().decode()


def conv_dims(F: int, Tin: int):
    d1, d2, t = i32(), i32(), i32()
    load().ds2_conv_dims(F, Tin, C.byref(d1), C.byref(d2), C.byref(t))
    return d1.value, d2.value, t.value
