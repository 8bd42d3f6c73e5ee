"""Typed torch-tensor wrappers over the libds2hip C-ABI (device pointers + current HIP stream).

PyTorch is plumbing here: it owns device memory (caching allocator), streams and RCCL; every FLOP of
the DeepSpeech2 step is executed by the hand-written gfx950 kernels behind `asr_amd/_lib.py`.
Nothing in this module has a CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional, Tuple

import torch

from . import _lib

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.DS2LibraryError("asr_amd kernels need GPU tensors (no CPU fallback exists)")
        if t.dtype != torch.float32:
            raise TypeError(f"expected float32 tensor, got {t.dtype}")


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


class RnnCtx(C.Structure):
    """include/ds2hip.h `ds2_rnn_ctx`: everything the recurrence entry points remember between calls, owned by the CALLER (the library
    keeps no mutable global state).  One per (thread, device): two Python threads driving two streams never see each other's kernel-path
    bits, cooldown, enable switches or starvation record."""
    _fields_ = [("size", C.c_int), ("persist_fwd", C.c_int), ("persist_bwd", C.c_int), ("cooldown", C.c_int), ("rearm_calls", C.c_int),
                ("starved_total", C.c_int), ("last_path", C.c_int), ("last_bwd_kind", C.c_int), ("debug_flags", C.c_int),
                ("ws_prearmed", C.c_int), ("reserved", C.c_int * 6), ("status_dev", C.c_void_p), ("poison_host", C.c_void_p), ("poison_dev", C.c_void_p)]


_RNN_TLS = threading.local()
_RNN_REGISTRY = []                      # every context ever created: (ctx, status tensor, pinned word, uid) — kept for the life of the process, so
_RNN_REGISTRY_LOCK = threading.Lock()   # a kernel still queued on some stream can never write into memory the caching allocator has handed on
_RNN_UID = {}                           # ctypes address of a context -> its monotonically increasing id (never reused, unlike thread idents)


def _pinned_word():
    """(tensor, host address, device address) of one pinned, mapped int32 word — or (None, None, None) when the runtime will not map it."""
    try:
        t = torch.zeros(16, dtype=torch.int32).pin_memory()
        hip = C.CDLL("libamdhip64.so")
        dptr = C.c_void_p()
        if hip.hipHostGetDevicePointer(C.byref(dptr), C.c_void_p(t.data_ptr()), 0) == 0 and dptr.value:
            return t, t.data_ptr(), dptr.value
    except (OSError, RuntimeError):
        pass
    return None, None, None


def _dev_key(device=None):
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return dev, (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())


def rnn_ctx(device=None) -> RnnCtx:
    """The recurrence context the calling thread uses on `device` (default: the current CUDA device): the one bound with use_rnn_ctx() if any
    (an autograd worker thread running the backward of a forward that another thread issued), else the thread's own, created on first use:
    the struct itself, its 8-int device status record and its pinned poison word are allocated HERE, by the caller of the C ABI."""
    dev, key = _dev_key(device)
    bound = getattr(_RNN_TLS, "bound", None)
    if bound and key in bound[-1]:
        return bound[-1][key]
    table = getattr(_RNN_TLS, "ctx", None)
    if table is None:
        table = _RNN_TLS.ctx = {}
    if key not in table:
        ctx = RnnCtx()
        status = torch.zeros(8, dtype=torch.int32, device=dev)
        pin, phost, pdev = _pinned_word()
        torch.cuda.synchronize(dev)                                   # the status record is zero before the first launch reads it
        _lib.check(_lib.load().ds2_rnn_ctx_init(C.addressof(ctx), status.data_ptr(), phost, pdev), "ds2_rnn_ctx_init")
        assert ctx.size == C.sizeof(RnnCtx), "ds2_rnn_ctx layout mismatch between include/ds2hip.h and asr_amd/ops.py"
        with _RNN_REGISTRY_LOCK:
            uid = len(_RNN_REGISTRY)
            _RNN_REGISTRY.append((ctx, status, pin, uid))
            _RNN_UID[C.addressof(ctx)] = uid
        table[key] = ctx
    return table[key]


class use_rnn_ctx:
    """`with use_rnn_ctx(ctx, device):` — inside the block the CALLING thread's recurrence calls on `device` go through `ctx` instead of the
    thread's own context.  asr_amd.modules.deepspeech binds the context of the thread that ran forward around engine.backward: `loss.backward()`
    runs on PyTorch's autograd worker thread, and the enable switches, debug selectors, starvation record and cooldown that the forward thread
    (the trainer) set and reads must be the ones the backward recurrences see and write."""

    def __init__(self, ctx: RnnCtx, device=None):
        self.ctx, self.key = ctx, _dev_key(device)[1]

    def __enter__(self):
        stack = getattr(_RNN_TLS, "bound", None)
        if stack is None:
            stack = _RNN_TLS.bound = []
        top = dict(stack[-1]) if stack else {}
        top[self.key] = self.ctx
        stack.append(top)
        return self.ctx

    def __exit__(self, *exc):
        _RNN_TLS.bound.pop()
        return False


def _ctxp(device=None) -> int:
    return C.addressof(rnn_ctx(device))


def rnn_ctx_key(device=None):
    """hashable identity of the context in use (host-side caches of "what did the last call of this shape do" are keyed with it): a
    monotonically increasing id, never reused by a later thread or context"""
    return _RNN_UID[_ctxp(device)]


def debug_flags(flags: int, device=None) -> int:
    """ds2_debug_flags on the calling thread's context: kernel-family selectors of the recurrence; returns the previous value."""
    return _lib.load().ds2_debug_flags(_ctxp(device), int(flags))


def _row_pitch(t: Tensor) -> int:
    """pitch (elements) of a 2-D row-major (possibly column-sliced) view."""
    assert t.dim() == 2 and t.stride(1) == 1, "need a row-major 2-D view"
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
def gemm_raw(transA: bool, transB: bool, M: int, N: int, K: int, A_ptr: int, lda: int, sA: int, B_ptr: int, ldb: int, sB: int,
             C_ptr: int, ldc: int, sC: int, device, bias: Optional[Tensor] = None, accumulate: bool = False, batch: int = 1,
             splitk: int = 0):
    lib = _lib.load()
    if splitk <= 0:
        tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
        splitk = 1
        if M <= 32 and K >= 4096:          # skinny weight gradient (the fc block: dW = dlogits^T Xn, 29 classes, K = T*B): a block's k-loop is
            # one HBM round trip per 16-deep tile with little else resident on its CU - more, shorter blocks hide it; the slabs are tiny
            splitk = max(1, min((1024 + tiles - 1) // tiles, K // 256))
        elif tiles < 400 and K >= 2048:      # under-filled grid with a long reduction: aim at >= 2 blocks per CU
            splitk = max(1, min((512 + tiles - 1) // tiles, K // 1024))
        elif tiles < 128 and K >= 1024:
            splitk = max(1, min((384 + tiles - 1) // tiles, K // 256))
    ws = None
    wsb = 0
    if splitk > 1:
        wsb = lib.ds2_gemm_f32_workspace_bytes(M, N, batch, splitk)
        ws = _ws(wsb, device)
    _lib.check(lib.ds2_gemm_f32(int(transA), int(transB), M, N, K, A_ptr, lda, sA, B_ptr, ldb, sB, C_ptr, ldc, sC, _ptr(bias),
                                int(accumulate), batch, splitk, _ptr(ws), wsb, _stream()), "ds2_gemm_f32")


def gemm(A: Tensor, B: Tensor, transA: bool = False, transB: bool = False, bias: Optional[Tensor] = None,
         out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """out[M,N] (+)= op(A) @ op(B) (+ bias).  A, B, out: 2-D row-major views (column slices allowed)."""
    _chk_f32(A, B, bias, out)
    M, K = (A.size(1), A.size(0)) if transA else (A.size(0), A.size(1))
    N = B.size(0) if transB else B.size(1)
    Kb = B.size(1) if transB else B.size(0)
    assert K == Kb, (A.shape, B.shape, transA, transB)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    gemm_raw(transA, transB, M, N, K, A.data_ptr(), _row_pitch(A), 0, B.data_ptr(), _row_pitch(B), 0, out.data_ptr(),
             _row_pitch(out), 0, A.device, bias, accumulate)
    return out


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def cast_bf16(src: Tensor, ld: Optional[int] = None) -> Tensor:
    """(R, C) fp32 row-major view -> (R, ld or pad8(C)) bf16, pad columns zero."""
    _chk_f32(src)
    R, Cc = src.shape
    ld = _pad8(Cc) if ld is None else ld
    assert ld >= Cc and ld % 8 == 0
    dst = torch.empty(R, ld, dtype=torch.bfloat16, device=src.device)
    _lib.check(_lib.load().ds2_cast_bf16(src.data_ptr(), _row_pitch(src), dst.data_ptr(), dst.size(1), R, Cc, _stream()), "ds2_cast_bf16")
    return dst


def cast_transpose_bf16(src: Tensor, colsum: Optional[Tensor] = None) -> Tensor:
    """(R, C) fp32 row-major view -> (C, pad8(R)) bf16 = src^T, pad columns zero.
    colsum: optional contiguous fp32 tensor of C elements that receives the column sums of src from the same read."""
    _chk_f32(src)
    R, Cc = src.shape
    lib = _lib.load()
    dst = torch.empty(Cc, _pad8(R), dtype=torch.bfloat16, device=src.device)
    if colsum is None:
        _lib.check(lib.ds2_cast_transpose_bf16(src.data_ptr(), _row_pitch(src), dst.data_ptr(), dst.size(1), R, Cc, _stream()),
                   "ds2_cast_transpose_bf16")
        return dst
    assert colsum.dtype == torch.float32 and colsum.is_cuda and colsum.is_contiguous() and colsum.numel() == Cc
    wsb = lib.ds2_cast_bf16_both_workspace_bytes(R, Cc)
    ws = _ws(wsb, src.device)
    _lib.check(lib.ds2_cast_bf16_both(src.data_ptr(), _row_pitch(src), 0, 0, dst.data_ptr(), dst.size(1), R, Cc, colsum.data_ptr(),
                                      ws.data_ptr(), wsb, _stream()), "ds2_cast_bf16_both")
    return dst


def cast_bf16_both(src: Tensor, colsum: bool = False, ld_r: Optional[int] = None):
    """(R, C) fp32 -> (bf16 (R, ld_r or pad8(C)), bf16 (C, pad8(R)) = src^T[, fp32 column sums (C)]) from one read of src; pads zero."""
    _chk_f32(src)
    R, Cc = src.shape
    lib = _lib.load()
    ld_r = _pad8(Cc) if ld_r is None else ld_r
    assert ld_r >= Cc and ld_r % 8 == 0
    dst_r = torch.empty(R, ld_r, dtype=torch.bfloat16, device=src.device)
    dst_t = torch.empty(Cc, _pad8(R), dtype=torch.bfloat16, device=src.device)
    cs, ws, wsb = None, None, 0
    if colsum:
        cs = torch.empty(Cc, dtype=torch.float32, device=src.device)
        wsb = lib.ds2_cast_bf16_both_workspace_bytes(R, Cc)
        ws = _ws(wsb, src.device)
    _lib.check(lib.ds2_cast_bf16_both(src.data_ptr(), _row_pitch(src), dst_r.data_ptr(), dst_r.size(1), dst_t.data_ptr(), dst_t.size(1),
                                      R, Cc, _ptr(cs), _ptr(ws), wsb, _stream()), "ds2_cast_bf16_both")
    return (dst_r, dst_t, cs) if colsum else (dst_r, dst_t)


def transpose_bf16(src: Tensor, colsum=False):
    """(R, C) bf16 row-major -> bf16 (C, pad8(R)) = src^T [, fp32 column sums (C)] from one read; pads zero.
    colsum: False | True (new tensor) | a contiguous fp32 tensor of C elements that receives the sums (e.g. a bias-gradient slice)."""
    assert src.dtype == torch.bfloat16 and src.is_cuda and src.dim() == 2 and src.stride(1) == 1 and src.stride(0) % 8 == 0
    R, Cc = src.shape
    lib = _lib.load()
    dst_t = torch.empty(Cc, _pad8(R), dtype=torch.bfloat16, device=src.device)
    cs, ws, wsb = None, None, 0
    if colsum is not False and colsum is not None:
        cs = torch.empty(Cc, dtype=torch.float32, device=src.device) if colsum is True else colsum
        assert cs.dtype == torch.float32 and cs.is_cuda and cs.is_contiguous() and cs.numel() == Cc
        wsb = lib.ds2_cast_bf16_both_workspace_bytes(R, Cc)
        ws = _ws(wsb, src.device)
    _lib.check(lib.ds2_transpose_bf16(src.data_ptr(), src.stride(0), dst_t.data_ptr(), dst_t.size(1), R, Cc, _ptr(cs), _ptr(ws), wsb,
                                      _stream()), "ds2_transpose_bf16")
    return (dst_t, cs) if cs is not None else dst_t


def _pick_splitk(M: int, N: int, K: int) -> int:
    """Split-K factor for the bf16 GEMM: trade chip fill (256 CUs, one 256x256 tile each per round) against the fp32 partial
    slabs a split writes and re-reads.  Rates are the measured ones (scripts/bench_gemm.py): ~1.1 PF/s per busy CU-round for the
    256-tile kernel, ~3 TB/s for slab traffic."""
    if M >= 256 and N >= 256:
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        best, best_cost = 1, None
        for s in (1, 2, 3, 4, 6, 8, 12, 16):
            if s > 1 and K // s < 1024:
                break
            rounds = -(-tiles * s // 256)
            util = tiles * s / (rounds * 256.0)
            cost = 2.0 * M * N * K / (1.1e15 * util) + (0.0 if s == 1 else (s + 1) * M * N * 4 / 3e12)
            if best_cost is None or cost < best_cost * 0.97:
                best, best_cost = s, cost
        return best
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles < 400 and K >= 4096:
        return max(1, min((768 + tiles - 1) // tiles, K // 2048))
    return 1


def gemm_bf16_nt_obf16(A: Tensor, B: Tensor, bias: Optional[Tensor] = None) -> Optional[Tensor]:
    """bf16 (M, N) = A[M,K] @ B[N,K]^T + bias with fp32 accumulation, ONE rounding at the store — or None where the four-wave kernel does not
    apply (the caller then takes gemm_bf16_nt's fp32 result)."""
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.is_cuda and B.is_cuda and A.stride(1) == 1 and B.stride(1) == 1
    M, K = A.shape
    N, Kb = B.shape
    assert K == Kb and K % 8 == 0, (A.shape, B.shape)
    if N % 8:
        return None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=A.device)
    rc = _lib.load().ds2_gemm_bf16_nt_obf16(M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), out.data_ptr(), N, _ptr(bias), _stream())
    if rc == 1:
        return None
    _lib.check(rc, "ds2_gemm_bf16_nt_obf16")
    return out


def widen_bf16(x: Tensor) -> Tensor:
    """contiguous bf16 -> fp32 (numel % 8 == 0)"""
    assert x.dtype == torch.bfloat16 and x.is_cuda and x.is_contiguous() and x.numel() % 8 == 0
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().ds2_cast_f32_from_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "ds2_cast_f32_from_bf16")
    return out


def gemm_bf16_nt(A: Tensor, B: Tensor, bias: Optional[Tensor] = None, out: Optional[Tensor] = None, accumulate: bool = False,
                 splitk: int = 0) -> Tensor:
    """out[M,N] fp32 (+)= A[M,K] @ B[N,K]^T, A/B bf16 with K (incl. zero padding) a multiple of 8."""
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.is_cuda and B.is_cuda and A.stride(1) == 1 and B.stride(1) == 1
    lib = _lib.load()
    M, K = A.shape
    N, Kb = B.shape
    assert K == Kb and K % 8 == 0, (A.shape, B.shape)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    if splitk <= 0:
        splitk = _pick_splitk(M, N, K)
    ws, wsb = None, 0
    if splitk > 1:
        wsb = lib.ds2_gemm_bf16_workspace_bytes(M, N, 1, splitk)
        ws = _ws(wsb, A.device)
    _lib.check(lib.ds2_gemm_bf16_nt(M, N, K, A.data_ptr(), A.stride(0), 0, B.data_ptr(), B.stride(0), 0, out.data_ptr(), _row_pitch(out), 0,
                                    _ptr(bias), int(accumulate), 1, splitk, _ptr(ws), wsb, _stream()), "ds2_gemm_bf16_nt")
    return out


def gemm_bf16_nt_pair(A0: Tensor, A1: Tensor, B0: Tensor, B1: Tensor, out: Tensor):
    """Two equal-shape NT products in ONE launch (batch = 2): out[d] (M, N) = A_d (M, K) @ B_d (N, K)^T.  A0/A1 (and B0/B1) must be
    views of one bf16 buffer with equal pitches; out (2, M, N) fp32 with equal-pitch slices.  Used for the two directions of dW_hh:
    twice the tiles per launch means half the split-K factor (and half the partial-slab traffic) for the same chip fill."""
    for t in (A0, A1, B0, B1):
        assert t.dtype == torch.bfloat16 and t.is_cuda and t.dim() == 2 and t.stride(1) == 1
    assert A0.shape == A1.shape and B0.shape == B1.shape and A0.stride(0) == A1.stride(0) and B0.stride(0) == B1.stride(0)
    M, K = A0.shape
    N = B0.size(0)
    assert B0.size(1) == K and K % 8 == 0 and out.dim() == 3 and out.size(0) == 2 and tuple(out.shape[1:]) == (M, N) and out.stride(2) == 1
    sA, sB = (A1.data_ptr() - A0.data_ptr()) // 2, (B1.data_ptr() - B0.data_ptr()) // 2
    assert (A1.data_ptr() - A0.data_ptr()) % 16 == 0 and (B1.data_ptr() - B0.data_ptr()) % 16 == 0
    lib = _lib.load()
    splitk = _pick_splitk(M, 2 * N, K)                 # chip fill is decided by the tiles of both products together
    ws, wsb = None, 0
    if splitk > 1:
        wsb = lib.ds2_gemm_bf16_workspace_bytes(M, N, 2, splitk)
        ws = _ws(wsb, A0.device)
    _lib.check(lib.ds2_gemm_bf16_nt(M, N, K, A0.data_ptr(), A0.stride(0), sA, B0.data_ptr(), B0.stride(0), sB, out.data_ptr(), out.stride(1),
                                    out.stride(0), None, 0, 2, splitk, _ptr(ws), wsb, _stream()), "ds2_gemm_bf16_nt")
    return out


def gemm_bf16_tn(A: Tensor, B: Tensor, out: Optional[Tensor] = None, accumulate: bool = False, splitk: int = 0) -> Tensor:
    """out[M,N] fp32 (+)= A[K,M]^T @ B[K,N]: A, B bf16 row-major views (column stride 1, any row pitch % 8 == 0) sharing the K rows."""
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.is_cuda and B.is_cuda and A.stride(1) == 1 and B.stride(1) == 1
    lib = _lib.load()
    K, M = A.shape
    Kb, N = B.shape
    assert K == Kb, (A.shape, B.shape)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    if splitk <= 0:
        splitk = _pick_splitk(M, N, K)
    ws, wsb = None, 0
    if splitk > 1:
        wsb = lib.ds2_gemm_bf16_workspace_bytes(M, N, 1, splitk)
        ws = _ws(wsb, A.device)
    _lib.check(lib.ds2_gemm_bf16_tn(M, N, K, A.data_ptr(), A.stride(0), 0, B.data_ptr(), B.stride(0), 0, out.data_ptr(), _row_pitch(out), 0,
                                    int(accumulate), 1, splitk, _ptr(ws), wsb, _stream()), "ds2_gemm_bf16_tn")
    return out


def gemm_bf16_tn_pair(A0: Tensor, A1: Tensor, B0: Tensor, B1: Tensor, out: Tensor, accumulate: bool = False):
    """Two equal-shape TN products in ONE launch: out[d] (M, N) = A_d (K, M)^T @ B_d (K, N).  A0/A1 (and B0/B1) are views of one bf16
    buffer with equal pitches (the two directions of dW_hh: column blocks of dGx / h at row offsets of +-B)."""
    for t in (A0, A1, B0, B1):
        assert t.dtype == torch.bfloat16 and t.is_cuda and t.dim() == 2 and t.stride(1) == 1
    assert A0.shape == A1.shape and B0.shape == B1.shape and A0.stride(0) == A1.stride(0) and B0.stride(0) == B1.stride(0)
    K, M = A0.shape
    N = B0.size(1)
    assert B0.size(0) == K and out.dim() == 3 and out.size(0) == 2 and tuple(out.shape[1:]) == (M, N) and out.stride(2) == 1
    dA, dB = A1.data_ptr() - A0.data_ptr(), B1.data_ptr() - B0.data_ptr()
    assert dA % 16 == 0 and dB % 16 == 0
    lib = _lib.load()
    splitk = _pick_splitk(M, 2 * N, K)
    ws, wsb = None, 0
    if splitk > 1:
        wsb = lib.ds2_gemm_bf16_workspace_bytes(M, N, 2, splitk)
        ws = _ws(wsb, A0.device)
    _lib.check(lib.ds2_gemm_bf16_tn(M, N, K, A0.data_ptr(), A0.stride(0), dA // 2, B0.data_ptr(), B0.stride(0), dB // 2, out.data_ptr(),
                                    out.stride(1), out.stride(0), int(accumulate), 2, splitk, _ptr(ws), wsb, _stream()), "ds2_gemm_bf16_tn")
    return out


def split_bf16(X: Tensor, order: int, pad_rows: int = 0) -> Tensor:
    """fp32 (R, C) row-major view -> its split bf16 operand (R, n * pad8(C)): x = hi + lo, hi = bf16(x), lo = bf16(x - hi); order 0 = [hi | hi | lo]
    (A operand of the fp32 mode's three-term product), 1 = [hi | lo | hi] (B operand), 2 = [hi | lo] (TN products).  See ds2_split_bf16.
    pad_rows: the result has its row count rounded up to a multiple of it, the extra rows ZERO — as the K-row-major operand of a TN product
    (reduction index = rows) it then has a reduction length the four-wave kernel takes (K % 64 == 0), and the zero rows add nothing."""
    _chk_f32(X)
    assert X.dim() == 2 and X.stride(1) == 1 and order in (0, 1, 2)
    R, Cc = X.shape
    Cp = (Cc + 7) // 8 * 8
    Rp = -(-R // pad_rows) * pad_rows if pad_rows else R
    out = torch.empty(Rp, (2 if order == 2 else 3) * Cp, dtype=torch.bfloat16, device=X.device)
    lib = _lib.load()
    if Rp > R:
        _lib.check(lib.ds2_memset_async(out[R:].data_ptr(), 0, (Rp - R) * out.stride(0) * 2, _stream()), "ds2_memset_async")
    _lib.check(lib.ds2_split_bf16(X.data_ptr(), _row_pitch(X), out.data_ptr(), out.stride(0), R, Cc, order, _stream()), "ds2_split_bf16")
    return out


def _tn_problem_array(problems):
    arr = (_lib.TnProblem * len(problems))()
    for q, (A, B, out) in zip(arr, problems):
        assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and out.dtype == torch.float32 and A.is_cuda and B.is_cuda and out.is_cuda
        assert A.dim() == 2 and B.dim() == 2 and out.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1 and out.stride(1) == 1
        K, M = A.shape
        assert B.size(0) == K and tuple(out.shape) == (M, B.size(1)), (A.shape, B.shape, out.shape)
        q.A, q.B, q.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
        q.M, q.N, q.K, q.lda, q.ldb, q.ldc = M, B.size(1), K, _row_pitch(A), _row_pitch(B), _row_pitch(out)
    return arr


def gemm_bf16_tn_splitk_group(problems, splitk: int = 0, epilogue=None):
    """Several TN products `[(A (K, M), B (K, N), out (M, N))]` in ONE launch of the 256 x 256 kernel with a common split-K factor and one reduce
    launch (ds2_gemm_bf16_tn_splitk_group).  splitk 0: chosen so that tiles x splitk fills whole rounds of the chip (the cost model of
    _pick_splitk over the tiles of all products together)."""
    lib = _lib.load()
    arr = _tn_problem_array(problems)
    if splitk <= 0:
        tiles = sum(((q.M + 255) // 256) * ((q.N + 255) // 256) for q in arr)
        kmin = min(q.K for q in arr)
        mn = sum(q.M * q.N for q in arr)
        best, best_cost = 1, None
        for s_ in (1, 2, 3, 4, 5, 6, 8, 12, 16):
            if s_ > 1 and kmin // s_ < 1024:
                break
            rounds = -(-tiles * s_ // 256)
            util = tiles * s_ / (rounds * 256.0)
            cost = 2.0 * mn * kmin / (1.1e15 * util) + (0.0 if s_ == 1 else (s_ + 1) * mn * 4 / 3e12)
            if best_cost is None or cost < best_cost * 0.97:
                best, best_cost = s_, cost
        splitk = best
    ap = C.cast(arr, C.c_void_p)
    wsb = lib.ds2_gemm_bf16_tn_splitk_group_workspace_bytes(len(problems), ap, splitk)
    ws = _ws(wsb, problems[0][0].device) if wsb else None
    if epilogue is not None:
        # epilogue = (index, scale (N), rowv (M), shift (N)): out[index] = (A^T B) diag(scale) + rowv (x) shift, applied by the reduce launch
        idx, scale, rowv, shift = epilogue
        _chk_f32(scale, rowv, shift)
        rc = lib.ds2_gemm_bf16_tn_splitk_group_ep(len(problems), ap, splitk, int(idx), scale.data_ptr(), rowv.data_ptr(), shift.data_ptr(), _ptr(ws), wsb, _stream())
        if rc == 0:
            return splitk
        if rc != 1:
            _lib.check(rc, "ds2_gemm_bf16_tn_splitk_group_ep")
        _lib.check(lib.ds2_gemm_bf16_tn_splitk_group(len(problems), ap, splitk, _ptr(ws), wsb, _stream()), "ds2_gemm_bf16_tn_splitk_group")
        scale_rank1_(problems[idx][2], scale, rowv, shift)       # (single slab: nothing reduces that product)
        return splitk
    _lib.check(lib.ds2_gemm_bf16_tn_splitk_group(len(problems), ap, splitk, _ptr(ws), wsb, _stream()), "ds2_gemm_bf16_tn_splitk_group")
    return splitk


def gemm_bf16_tn_group(problems, max_workgroups: int = 0):
    """Several TN products in ONE launch of the co-resident kernel (csrc/gemm_tn_group.h): `problems` = [(A (K, M), B (K, N), out (M, N))],
    A / B bf16 row-major views sharing their K rows, out fp32 (row pitch arbitrary, columns contiguous).  out = A^T @ B, every tile a full
    reduction over K (no split-K, no workspace)."""
    lib = _lib.load()
    arr = (_lib.TnProblem * len(problems))()
    for q, (A, B, out) in zip(arr, problems):
        assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and out.dtype == torch.float32 and A.is_cuda and B.is_cuda and out.is_cuda
        assert A.dim() == 2 and B.dim() == 2 and out.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1 and out.stride(1) == 1
        K, M = A.shape
        assert B.size(0) == K and tuple(out.shape) == (M, B.size(1)), (A.shape, B.shape, out.shape)
        q.A, q.B, q.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
        q.M, q.N, q.K, q.lda, q.ldb, q.ldc = M, B.size(1), K, _row_pitch(A), _row_pitch(B), _row_pitch(out)
    _lib.check(lib.ds2_gemm_bf16_tn_group(len(problems), C.cast(arr, C.c_void_p), int(max_workgroups), _stream()), "ds2_gemm_bf16_tn_group")


# ------------------------------------------------------------------------------------------------
# BatchNorm1d family on (M, H)
# ------------------------------------------------------------------------------------------------
def colstats(X: Tensor, run_mean: Optional[Tensor] = None, run_var: Optional[Tensor] = None):
    _chk_f32(X, run_mean, run_var)
    lib = _lib.load()
    M, H = X.shape
    mean = torch.empty(H, dtype=torch.float32, device=X.device)
    var = torch.empty_like(mean)
    wsb = lib.ds2_colreduce_workspace_bytes(M, H)
    ws = _ws(wsb, X.device)
    _lib.check(lib.ds2_colstats_f32(X.data_ptr(), _row_pitch(X), M, H, mean.data_ptr(), var.data_ptr(), _ptr(run_mean),
                                    _ptr(run_var), BN_MOMENTUM, ws.data_ptr(), wsb, _stream()), "ds2_colstats_f32")
    return mean, var


def add_colstats(Xa: Tensor, Xb: Tensor, run_mean: Optional[Tensor] = None, run_var: Optional[Tensor] = None):
    """Y = Xa + Xb with column mean / biased var of Y (and optional running-stat update)."""
    _chk_f32(Xa, Xb, run_mean, run_var)
    lib = _lib.load()
    M, H = Xa.shape
    Y = torch.empty(M, H, dtype=torch.float32, device=Xa.device)
    mean = torch.empty(H, dtype=torch.float32, device=Xa.device)
    var = torch.empty_like(mean)
    wsb = lib.ds2_colreduce_workspace_bytes(M, H)
    ws = _ws(wsb, Xa.device)
    _lib.check(lib.ds2_add_colstats_f32(Xa.data_ptr(), _row_pitch(Xa), Xb.data_ptr(), _row_pitch(Xb), Y.data_ptr(), H, M, H,
                                        mean.data_ptr(), var.data_ptr(), _ptr(run_mean), _ptr(run_var), BN_MOMENTUM,
                                        ws.data_ptr(), wsb, _stream()), "ds2_add_colstats_f32")
    return Y, mean, var


def center_colstats(Xa: Tensor, Xb: Tensor, hsum: Tensor, run_mean: Optional[Tensor] = None, run_var: Optional[Tensor] = None):
    """(yc bf16 (M, pad8(H)) = (Xa + Xb) - m0 with m0 the column means from `hsum` (2, tiles, H; rnn_fwd), mean, var, delta): the direction
    sum and the BatchNorm1d statistics of a recurrent layer's output in ONE pass that writes only the centred bf16 GEMM operand."""
    _chk_f32(Xa, Xb, hsum, run_mean, run_var)
    lib = _lib.load()
    M, H = Xa.shape
    assert hsum.is_contiguous() and hsum.dim() == 3 and hsum.size(0) == 2 and hsum.size(2) == H
    yc = torch.empty(M, _pad8(H), dtype=torch.bfloat16, device=Xa.device)
    stats = torch.empty(3, H, dtype=torch.float32, device=Xa.device)
    wsb = lib.ds2_colreduce_workspace_bytes(M, H) + 4 * H
    ws = _ws(wsb, Xa.device)
    _lib.check(lib.ds2_center_colstats(Xa.data_ptr(), _row_pitch(Xa), Xb.data_ptr(), _row_pitch(Xb), hsum.data_ptr(), hsum.size(1), yc.data_ptr(), yc.size(1),
                                       M, H, stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), _ptr(run_mean), _ptr(run_var), BN_MOMENTUM,
                                       ws.data_ptr(), wsb, _stream()), "ds2_center_colstats")
    return yc, stats[0], stats[1], stats[2]


def wih_fold(W: Tensor, bias: Tensor, var: Tensor, gamma: Tensor, beta: Tensor, delta: Tensor, ld: int):
    """BatchNorm1d folded into the projection behind it: (W2 bf16 (R, ld) = W diag(s), bias2 fp32 = bias + W c, s, c) with
    s = gamma rsqrt(var + eps), c = beta - delta s  (ds2_wih_fold_bf16)."""
    _chk_f32(W, bias, var, gamma, beta, delta)
    R, I = W.shape
    assert ld >= I and ld % 8 == 0 and I % 4 == 0 and W.stride(1) == 1
    W2 = torch.empty(R, ld, dtype=torch.bfloat16, device=W.device)
    bias2 = torch.empty(R, dtype=torch.float32, device=W.device)
    sc = torch.empty(2, I, dtype=torch.float32, device=W.device)
    _lib.check(_lib.load().ds2_wih_fold_bf16(W.data_ptr(), _row_pitch(W), bias.data_ptr(), R, I, var.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             delta.data_ptr(), BN_EPS, W2.data_ptr(), ld, bias2.data_ptr(), sc[0].data_ptr(), sc[1].data_ptr(), _stream()),
               "ds2_wih_fold_bf16")
    return W2, bias2, sc[0], sc[1]


def scale_rank1_(Cm: Tensor, scale: Tensor, rowv: Tensor, shift: Tensor) -> Tensor:
    """in place: C[r][c] = C[r][c] * scale[c] + rowv[r] * shift[c]"""
    _chk_f32(Cm, scale, rowv, shift)
    R, N = Cm.shape
    assert Cm.stride(1) == 1 and scale.numel() == N == shift.numel() and rowv.numel() == R and rowv.is_contiguous()
    _lib.check(_lib.load().ds2_scale_rank1_f32(Cm.data_ptr(), _row_pitch(Cm), R, N, scale.data_ptr(), rowv.data_ptr(), shift.data_ptr(), _stream()),
               "ds2_scale_rank1_f32")
    return Cm


def bn1d_bwd_sums_xbf(dY: Tensor, X: Tensor, mean: Tensor, var: Tensor, gamma: Tensor, out: Optional[Tuple[Tensor, Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """bn1d_bwd_sums with the BatchNorm input X as bf16 (M, >= H): (sum dY, sum dY * xhat) with xhat = (X - mean) rsqrt(var + eps)."""
    _chk_f32(dY, mean, var, gamma)
    assert X.dtype == torch.bfloat16 and X.is_cuda and X.stride(1) == 1
    lib = _lib.load()
    M, H = dY.shape
    if out is None:
        both = torch.empty(2, H, dtype=torch.float32, device=dY.device)
        out = (both[0], both[1])
    s0, s1 = out
    _chk_f32(s0, s1)
    assert s0.is_contiguous() and s1.is_contiguous() and s0.numel() == H == s1.numel()
    wsb = lib.ds2_colreduce_workspace_bytes(M, H)
    ws = _ws(wsb, dY.device)
    _lib.check(lib.ds2_bn1d_bwd_xbf16(dY.data_ptr(), _row_pitch(dY), X.data_ptr(), X.stride(0), None, 0, M, H, mean.data_ptr(), var.data_ptr(),
                                      gamma.data_ptr(), BN_EPS, s1.data_ptr(), s0.data_ptr(), ws.data_ptr(), wsb, _stream()), "ds2_bn1d_bwd_xbf16")
    return s0, s1


def bn1d_bwd_xbf(dY: Tensor, X: Tensor, mean: Tensor, var: Tensor, gamma: Tensor, dgamma: Tensor, dbeta: Tensor) -> Tensor:
    """materialised BatchNorm1d backward with the input as bf16 (the fallback of layers whose recurrence cannot fuse it)"""
    _chk_f32(dY, mean, var, gamma, dgamma, dbeta)
    assert X.dtype == torch.bfloat16 and X.is_cuda and X.stride(1) == 1
    lib = _lib.load()
    M, H = dY.shape
    dX = torch.empty(M, H, dtype=torch.float32, device=dY.device)
    wsb = lib.ds2_colreduce_workspace_bytes(M, H)
    ws = _ws(wsb, dY.device)
    _lib.check(lib.ds2_bn1d_bwd_xbf16(dY.data_ptr(), _row_pitch(dY), X.data_ptr(), X.stride(0), dX.data_ptr(), H, M, H, mean.data_ptr(), var.data_ptr(),
                                      gamma.data_ptr(), BN_EPS, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), wsb, _stream()), "ds2_bn1d_bwd_xbf16")
    return dX


def colsum(X: Tensor) -> Tensor:
    _chk_f32(X)
    lib = _lib.load()
    M, H = X.shape
    s = torch.empty(H, dtype=torch.float32, device=X.device)
    s2 = torch.empty_like(s)
    wsb = lib.ds2_colreduce_workspace_bytes(M, H)
    ws = _ws(wsb, X.device)
    _lib.check(lib.ds2_colsum_f32(X.data_ptr(), _row_pitch(X), M, H, s.data_ptr(), s2.data_ptr(), ws.data_ptr(), wsb, _stream()),
               "ds2_colsum_f32")
    return s


def bn1d_apply(X: Tensor, mean: Tensor, var: Tensor, gamma: Tensor, beta: Tensor) -> Tensor:
    _chk_f32(X, mean, var, gamma, beta)
    M, H = X.shape
    Y = torch.empty(M, H, dtype=torch.float32, device=X.device)
    _lib.check(_lib.load().ds2_bn1d_apply_f32(X.data_ptr(), _row_pitch(X), Y.data_ptr(), H, M, H, mean.data_ptr(), var.data_ptr(),
                                              gamma.data_ptr(), beta.data_ptr(), BN_EPS, _stream()), "ds2_bn1d_apply_f32")
    return Y


def bn1d_apply_bf16(X: Tensor, mean: Tensor, var: Tensor, gamma: Tensor, beta: Tensor) -> Tensor:
    """BN(X) written straight as bf16 (M, pad8(H)), pad columns zero."""
    _chk_f32(X, mean, var, gamma, beta)
    M, H = X.shape
    Y = torch.empty(M, _pad8(H), dtype=torch.bfloat16, device=X.device)
    _lib.check(_lib.load().ds2_bn1d_apply_bf16(X.data_ptr(), _row_pitch(X), Y.data_ptr(), Y.size(1), M, H, mean.data_ptr(), var.data_ptr(),
                                               gamma.data_ptr(), beta.data_ptr(), BN_EPS, _stream()), "ds2_bn1d_apply_bf16")
    return Y


def bn1d_bwd(dY: Tensor, X: Tensor, mean: Tensor, var: Tensor, gamma: Tensor, dgamma: Tensor, dbeta: Tensor,
             dX: Optional[Tensor] = None) -> Tensor:
    _chk_f32(dY, X, mean, var, gamma, dgamma, dbeta, dX)
    lib = _lib.load()
    M, H = X.shape
    if dX is None:
        dX = torch.empty(M, H, dtype=torch.float32, device=X.device)
    wsb = lib.ds2_colreduce_workspace_bytes(M, H)
    ws = _ws(wsb, X.device)
    _lib.check(lib.ds2_bn1d_bwd_f32(dY.data_ptr(), _row_pitch(dY), X.data_ptr(), _row_pitch(X), dX.data_ptr(), _row_pitch(dX), M, H,
                                    mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), BN_EPS, dgamma.data_ptr(), dbeta.data_ptr(),
                                    ws.data_ptr(), wsb, _stream()), "ds2_bn1d_bwd_f32")
    return dX


# ------------------------------------------------------------------------------------------------
# BatchNorm2d + Hardtanh + mask on (B, C, D, T)
# ------------------------------------------------------------------------------------------------
def bn2d_stats(Y: Tensor, run_mean: Optional[Tensor] = None, run_var: Optional[Tensor] = None):
    _chk_f32(Y, run_mean, run_var)
    lib = _lib.load()
    B, Cc, D, T = Y.shape
    mean = torch.empty(Cc, dtype=torch.float32, device=Y.device)
    var = torch.empty_like(mean)
    wsb = lib.ds2_chanreduce_workspace_bytes(Cc)
    ws = _ws(wsb, Y.device)
    _lib.check(lib.ds2_bn2d_stats_f32(Y.data_ptr(), B, Cc, D, T, mean.data_ptr(), var.data_ptr(), _ptr(run_mean), _ptr(run_var),
                                      BN_MOMENTUM, ws.data_ptr(), wsb, _stream()), "ds2_bn2d_stats_f32")
    return mean, var


def bn2d_act_fwd(Y: Tensor, lens_dev: Tensor, mean, var, gamma, beta) -> Tensor:
    _chk_f32(Y, mean, var, gamma, beta)
    B, Cc, D, T = Y.shape
    A = torch.empty_like(Y)
    _lib.check(_lib.load().ds2_bn2d_act_fwd_f32(Y.data_ptr(), A.data_ptr(), B, Cc, D, T, lens_dev.data_ptr(), mean.data_ptr(),
                                                var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), BN_EPS, _stream()),
               "ds2_bn2d_act_fwd_f32")
    return A


def bn2d_act_bwd(Y: Tensor, dA: Tensor, lens_dev: Tensor, mean, var, gamma, beta, dgamma: Tensor, dbeta: Tensor) -> Tensor:
    _chk_f32(Y, dA, mean, var, gamma, beta, dgamma, dbeta)
    lib = _lib.load()
    B, Cc, D, T = Y.shape
    dY = torch.empty_like(Y)
    wsb = lib.ds2_chanreduce_workspace_bytes(Cc)
    ws = _ws(wsb, Y.device)
    _lib.check(lib.ds2_bn2d_act_bwd_f32(Y.data_ptr(), dA.data_ptr(), dY.data_ptr(), B, Cc, D, T, lens_dev.data_ptr(), mean.data_ptr(),
                                        var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), BN_EPS, dgamma.data_ptr(), dbeta.data_ptr(),
                                        ws.data_ptr(), wsb, _stream()), "ds2_bn2d_act_bwd_f32")
    return dY


def bn2d_act_fwd_fused(Y: Tensor, lens_dev: Tensor, mean, var, gamma, beta, want_f32=False, want_pad=False, want_nhwc=False):
    """bf16 mode: BatchNorm2d + Hardtanh + mask with the layout casts fused (csrc/norm.hip: bn2d_tile_kernel).  Returns
    (a_f32 (B,32,D,T) | None, a_pad (B,32,D,Tp) bf16 | None, a_nhwc (B,D,T,32) bf16 | None)."""
    _chk_f32(Y, mean, var, gamma, beta)
    lib = _lib.load()
    B, Cc, D, T = Y.shape
    assert Cc == 32 and Y.is_contiguous() and (want_f32 or want_pad or want_nhwc)
    a32 = torch.empty_like(Y) if want_f32 else None
    apad = torch.empty(B, 32, D, lib.ds2_conv_padded_pitch(T), dtype=torch.bfloat16, device=Y.device) if want_pad else None
    anh = torch.empty(B, D, T, 32, dtype=torch.bfloat16, device=Y.device) if want_nhwc else None
    _lib.check(lib.ds2_bn2d_act_fwd_fused(Y.data_ptr(), B, D, T, lens_dev.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(),
                                          beta.data_ptr(), BN_EPS, _ptr(a32), _ptr(apad), _ptr(anh), _stream()), "ds2_bn2d_act_fwd_fused")
    return a32, apad, anh


def bn2d_act_collapse(Y: Tensor, lens_dev: Tensor, mean, var, gamma, beta, want_f32=False, want_bf16=True, pad_to: int = 8):
    """bf16 mode: BatchNorm2d + Hardtanh + mask + (B,32*D,T) -> (T*B, 32*D) collapse (+ cast) in one pass over the conv output.
    Returns (x fp32 (T*B, 32*D) | None, x bf16 (T*B, 32*D rounded up to a multiple of pad_to, pad columns zero) | None)."""
    _chk_f32(Y, mean, var, gamma, beta)
    B, Cc, D, T = Y.shape
    assert Cc == 32 and Y.is_contiguous() and (want_f32 or want_bf16)
    F = 32 * D
    x32 = torch.empty(T * B, F, dtype=torch.float32, device=Y.device) if want_f32 else None
    xbf = None
    if want_bf16:
        xbf = torch.empty(T * B, -(-F // pad_to) * pad_to, dtype=torch.bfloat16, device=Y.device)      # (the kernel writes the pad columns)
    _lib.check(_lib.load().ds2_bn2d_act_collapse(Y.data_ptr(), B, D, T, lens_dev.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(),
                                                 beta.data_ptr(), BN_EPS, _ptr(x32), _ptr(xbf), xbf.size(1) if xbf is not None else 0, _stream()),
               "ds2_bn2d_act_collapse")
    return x32, xbf


def bn2d_act_bwd_fused(Y: Tensor, dA: Tensor, lens_dev: Tensor, mean, var, gamma, beta, dgamma: Tensor, dbeta: Tensor, dbias: Tensor,
                       want_f32=False, want_pad=False, want_nhwc=False):
    """bf16 mode: backward of the same block; writes dgamma / dbeta / dbias (the bias gradient of the convolution in front) in place and
    returns dY as (fp32 | None, zero-padded bf16 | None, channels-last bf16 | None)."""
    _chk_f32(Y, dA, mean, var, gamma, beta, dgamma, dbeta, dbias)
    lib = _lib.load()
    B, Cc, D, T = Y.shape
    assert Cc == 32 and Y.is_contiguous() and dA.is_contiguous() and dA.shape == Y.shape
    d32 = torch.empty_like(Y) if want_f32 else None
    dpad = torch.empty(B, 32, D, lib.ds2_conv_padded_pitch(T), dtype=torch.bfloat16, device=Y.device) if want_pad else None
    dnh = torch.empty(B, D, T, 32, dtype=torch.bfloat16, device=Y.device) if want_nhwc else None
    wsb = lib.ds2_bn2d_act_bwd_fused_workspace_bytes(B, D, T)
    ws = _ws(wsb, Y.device)
    _lib.check(lib.ds2_bn2d_act_bwd_fused(Y.data_ptr(), dA.data_ptr(), B, D, T, lens_dev.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                          gamma.data_ptr(), beta.data_ptr(), BN_EPS, dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(),
                                          _ptr(d32), _ptr(dpad), _ptr(dnh), ws.data_ptr(), wsb, _stream()), "ds2_bn2d_act_bwd_fused")
    return d32, dpad, dnh


def chan_sum(Y: Tensor) -> Tensor:
    """per-channel sum over (B, D, T) of a (B,C,D,T) tensor (bias gradients)."""
    mean, _ = bn2d_stats(Y)
    B, Cc, D, T = Y.shape
    return mean * float(B * D * T)


def transpose_bft(src: Tensor, B: int, F: int, T: int, to_tbf: bool) -> Tensor:
    _chk_f32(src)
    dst = torch.empty((T, B, F) if to_tbf else (B, F, T), dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().ds2_transpose_bft_f32(src.data_ptr(), dst.data_ptr(), B, F, T, 0 if to_tbf else 1, _stream()),
               "ds2_transpose_bft_f32")
    return dst


def transpose_batched(src: Tensor) -> Tensor:
    """(batch, R, C) contiguous -> (batch, C, R) contiguous."""
    _chk_f32(src)
    nb, R, Cc = src.shape
    dst = torch.empty(nb, Cc, R, dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().ds2_transpose2d_f32(src.data_ptr(), Cc, R * Cc, dst.data_ptr(), R, R * Cc, R, Cc, nb, _stream()),
               "ds2_transpose2d_f32")
    return dst


# ------------------------------------------------------------------------------------------------
# conv front-end
# ------------------------------------------------------------------------------------------------
def conv_pack(w1: Tensor, w2: Tensor):
    _chk_f32(w1, w2)
    lib = _lib.load()
    dev = w1.device
    wpk1 = torch.empty(lib.ds2_conv_packed_floats(0), dtype=torch.float32, device=dev)
    wpk2 = torch.empty(lib.ds2_conv_packed_floats(1), dtype=torch.float32, device=dev)
    wpk2d = torch.empty(lib.ds2_conv_packed_floats(2), dtype=torch.float32, device=dev)
    _lib.check(lib.ds2_conv_pack_f32(w1.data_ptr(), w2.data_ptr(), wpk1.data_ptr(), wpk2.data_ptr(), wpk2d.data_ptr(), _stream()),
               "ds2_conv_pack_f32")
    return wpk1, wpk2, wpk2d


def conv1_fwd(x: Tensor, wpk1: Tensor, bias: Tensor, lens_dev: Tensor) -> Tensor:
    _chk_f32(x, wpk1, bias)
    B, _, F, Tin = x.shape
    D1, D2, T = _lib.conv_dims(F, Tin)
    y1 = torch.empty(B, 32, D1, T, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().ds2_conv1_fwd_f32(x.data_ptr(), wpk1.data_ptr(), bias.data_ptr(), lens_dev.data_ptr(), y1.data_ptr(), B, F,
                                             Tin, _stream()), "ds2_conv1_fwd_f32")
    return y1


def conv2_fwd(a1: Tensor, wpk2: Tensor, bias: Tensor, lens_dev: Tensor) -> Tensor:
    _chk_f32(a1, wpk2, bias)
    B, _, D1, T = a1.shape
    D2 = (D1 + 20 - 21) // 2 + 1
    y2 = torch.empty(B, 32, D2, T, dtype=torch.float32, device=a1.device)
    _lib.check(_lib.load().ds2_conv2_fwd_f32(a1.data_ptr(), wpk2.data_ptr(), bias.data_ptr(), lens_dev.data_ptr(), y2.data_ptr(), B, D1,
                                             T, _stream()), "ds2_conv2_fwd_f32")
    return y2


def conv2_dgrad(dy2: Tensor, wpk2d: Tensor, D1: int) -> Tensor:
    _chk_f32(dy2, wpk2d)
    B, _, D2, T = dy2.shape
    da1 = torch.empty(B, 32, D1, T, dtype=torch.float32, device=dy2.device)
    _lib.check(_lib.load().ds2_conv2_dgrad_f32(dy2.data_ptr(), wpk2d.data_ptr(), da1.data_ptr(), B, D1, T, _stream()),
               "ds2_conv2_dgrad_f32")
    return da1


def conv1_wgrad(x: Tensor, dy1: Tensor, lens_dev: Tensor, dW1: Tensor):
    _chk_f32(x, dy1, dW1)
    lib = _lib.load()
    B, _, F, Tin = x.shape
    wsb = lib.ds2_conv_wgrad_workspace_bytes(0, B, F)
    ws = _ws(wsb, x.device)
    _lib.check(lib.ds2_conv1_wgrad_f32(x.data_ptr(), dy1.data_ptr(), lens_dev.data_ptr(), dW1.data_ptr(), B, F, Tin, 0, ws.data_ptr(),
                                       wsb, _stream()), "ds2_conv1_wgrad_f32")


def conv2_wgrad(a1: Tensor, dy2: Tensor, lens_dev: Tensor, dW2: Tensor):
    _chk_f32(a1, dy2, dW2)
    lib = _lib.load()
    B, _, D1, T = a1.shape
    wsb = lib.ds2_conv_wgrad_workspace_bytes(1, B, 161)
    ws = _ws(wsb, a1.device)
    _lib.check(lib.ds2_conv2_wgrad_f32(a1.data_ptr(), dy2.data_ptr(), lens_dev.data_ptr(), dW2.data_ptr(), B, D1, T, 0, ws.data_ptr(),
                                       wsb, _stream()), "ds2_conv2_wgrad_f32")


def conv1_pack_bf16(w1: Tensor) -> Tensor:
    _chk_f32(w1)
    lib = _lib.load()
    wp = torch.empty(lib.ds2_conv1_bf16_bytes(0, 0, 0, 0), dtype=torch.uint8, device=w1.device)
    _lib.check(lib.ds2_conv1_pack_bf16(w1.data_ptr(), wp.data_ptr(), _stream()), "ds2_conv1_pack_bf16")
    return wp


def conv1_gather_bf16(x: Tensor, want_fwd: bool = True, want_wgrad: bool = True):
    """x (B,1,F,Tin) fp32 -> (XB (B,F,P) bf16 rows, XB[..., 7 + s] = x[..., s], zero padded | None, X16T (B,F,16,pad64(T)) bf16 | None):
    conv1's forward / weight-gradient operand images."""
    _chk_f32(x)
    assert x.is_contiguous()
    lib = _lib.load()
    B, _, F, Tin = x.shape
    _, _, T = _lib.conv_dims(F, Tin)
    XB = torch.empty(lib.ds2_conv1_bf16_bytes(1, B, F, T) // 2, dtype=torch.bfloat16, device=x.device).view(B, F, lib.ds2_conv1_bf16_row_pitch(T)) if want_fwd else None
    X16T = torch.empty(lib.ds2_conv1_bf16_bytes(2, B, F, T) // 2, dtype=torch.bfloat16, device=x.device).view(B, F, 16, -1) if want_wgrad else None
    _lib.check(lib.ds2_conv1_gather_bf16(x.data_ptr(), _ptr(XB), _ptr(X16T), B, F, Tin, _stream()), "ds2_conv1_gather_bf16")
    return XB, X16T


def conv1_fwd_bf16(XB: Tensor, wp: Tensor, bias: Tensor, lens_dev: Tensor, Tin: int, stats: bool = False):
    """stats=True: also returns the per-block (sum, sum of squares) partials of y1's channels, taken in the epilogue (chanstats_from_partials)."""
    B, F, P = XB.shape
    D1, _, T = _lib.conv_dims(F, Tin)
    lib = _lib.load()
    assert P == lib.ds2_conv1_bf16_row_pitch(T) and XB.is_contiguous()
    y1 = torch.empty(B, 32, D1, T, dtype=torch.float32, device=XB.device)
    part = torch.empty(lib.ds2_conv1_fwd_bf16_stat_blocks(B, F, Tin), 32, 2, dtype=torch.float32, device=XB.device) if stats else None
    _lib.check(lib.ds2_conv1_fwd_bf16_stats(XB.data_ptr(), wp.data_ptr(), bias.data_ptr(), lens_dev.data_ptr(), y1.data_ptr(), B, F, Tin,
                                            _ptr(part), _stream()), "ds2_conv1_fwd_bf16")
    return (y1, part) if stats else y1


def chanstats_from_partials(part: Tensor, count: int, run_mean: Optional[Tensor] = None, run_var: Optional[Tensor] = None):
    """(nblk, C, 2) per-block channel sums of a conv forward epilogue -> (mean, biased var) over `count` elements per channel (+ running stats)."""
    _chk_f32(part, run_mean, run_var)
    nblk, Cc, _ = part.shape
    mean = torch.empty(Cc, dtype=torch.float32, device=part.device)
    var = torch.empty_like(mean)
    lib = _lib.load()
    wsb = lib.ds2_chanstats_from_partials_workspace_bytes()
    ws = _ws(wsb, part.device)
    _lib.check(lib.ds2_chanstats_from_partials(part.data_ptr(), nblk, Cc, float(count), mean.data_ptr(), var.data_ptr(), _ptr(run_mean),
                                               _ptr(run_var), BN_MOMENTUM, ws.data_ptr(), wsb, _stream()), "ds2_chanstats_from_partials")
    return mean, var


def conv1_wgrad_bf16(X16T: Tensor, dy1: Tensor, lens_dev: Tensor, dW1: Tensor, Tin: int):
    _chk_f32(dy1, dW1)
    lib = _lib.load()
    B, F = X16T.shape[0], X16T.shape[1]
    wsb = lib.ds2_conv1_wgrad_bf16_workspace_bytes(B, Tin)
    ws = _ws(wsb, dy1.device)
    _lib.check(lib.ds2_conv1_wgrad_bf16(X16T.data_ptr(), dy1.data_ptr(), lens_dev.data_ptr(), dW1.data_ptr(), B, F, Tin, ws.data_ptr(), wsb,
                                        _stream()), "ds2_conv1_wgrad_bf16")


def conv2_pack_bf16(w2: Tensor):
    _chk_f32(w2)
    lib = _lib.load()
    bufs = [torch.empty(lib.ds2_conv2_bf16_packed_bytes(i), dtype=torch.uint8, device=w2.device) for i in range(3)]
    _lib.check(lib.ds2_conv2_pack_bf16(w2.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), _stream()),
               "ds2_conv2_pack_bf16")
    return tuple(bufs)


def nhwc_bf16(x: Tensor) -> Tensor:
    """(B,32,D,T) fp32 -> (B,D,T,32) bf16 channels-last."""
    _chk_f32(x)
    B, Cc, D, T = x.shape
    assert Cc == 32 and x.is_contiguous()
    out = torch.empty(B, D, T, 32, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().ds2_nhwc_bf16_f32(x.data_ptr(), out.data_ptr(), B, D, T, _stream()), "ds2_nhwc_bf16_f32")
    return out


def bf16_residual(x: Tensor) -> Tensor:
    """x - float(bf16(x)): what a bf16 operand of x drops (the "lo" term of the fp32 mode's split products on the conv stack)."""
    _chk_f32(x)
    assert x.is_contiguous()
    r = torch.empty_like(x)
    _lib.check(_lib.load().ds2_bf16_residual_f32(x.data_ptr(), r.data_ptr(), x.numel(), _stream()), "ds2_bf16_residual_f32")
    return r


def sum3_(a: Tensor, b: Tensor, c: Optional[Tensor] = None) -> Tensor:
    """a += b (+ c) in place (the partial results of a split product)."""
    _chk_f32(a, b) if c is None else _chk_f32(a, b, c)
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and (c is None or (c.is_contiguous() and c.shape == a.shape))
    _lib.check(_lib.load().ds2_sum3_f32(a.data_ptr(), b.data_ptr(), _ptr(c), a.data_ptr(), a.numel(), _stream()), "ds2_sum3_f32")
    return a


def conv2_fwd_bf16(a1_nhwc: Tensor, wf: Tensor, bias: Tensor, lens_dev: Tensor, stats: bool = False):
    B, D1, T, _ = a1_nhwc.shape
    D2 = (D1 + 20 - 21) // 2 + 1
    lib = _lib.load()
    y2 = torch.empty(B, 32, D2, T, dtype=torch.float32, device=a1_nhwc.device)
    part = torch.empty(lib.ds2_conv2_fwd_bf16_stat_blocks(B, D1, T), 32, 2, dtype=torch.float32, device=a1_nhwc.device) if stats else None
    _lib.check(lib.ds2_conv2_fwd_bf16_stats(a1_nhwc.data_ptr(), wf.data_ptr(), bias.data_ptr(), lens_dev.data_ptr(), y2.data_ptr(), B, D1, T,
                                            _ptr(part), _stream()), "ds2_conv2_fwd_bf16")
    return (y2, part) if stats else y2


def conv2_dgrad_bf16(dy2_nhwc: Tensor, wd0: Tensor, wd1: Tensor, D1: int) -> Tensor:
    B, D2, T, _ = dy2_nhwc.shape
    da1 = torch.empty(B, 32, D1, T, dtype=torch.float32, device=dy2_nhwc.device)
    _lib.check(_lib.load().ds2_conv2_dgrad_bf16(dy2_nhwc.data_ptr(), wd0.data_ptr(), wd1.data_ptr(), da1.data_ptr(), B, D1, T, _stream()),
               "ds2_conv2_dgrad_bf16")
    return da1


def padcast_bf16(x: Tensor) -> Tensor:
    """(B,C,D,T) fp32 -> (B,C,D,Tp) bf16 with 8 leading zeros per row and a zero tail (operands of conv2_wgrad_bf16)."""
    _chk_f32(x)
    assert x.is_contiguous()
    lib = _lib.load()
    B, Cc, D, T = x.shape
    Tp = lib.ds2_conv_padded_pitch(T)
    out = torch.empty(B, Cc, D, Tp, dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.ds2_padcast_bf16(x.data_ptr(), out.data_ptr(), B * Cc * D, T, _stream()), "ds2_padcast_bf16")
    return out


def conv2_wgrad_bf16(a1p: Tensor, dy2p: Tensor, lens_dev: Tensor, dW2: Tensor, T: int):
    lib = _lib.load()
    B, _, D1, _ = a1p.shape
    wsb = lib.ds2_conv2_wgrad_bf16_workspace_bytes(B, D1)
    ws = _ws(wsb, a1p.device)
    _lib.check(lib.ds2_conv2_wgrad_bf16(a1p.data_ptr(), dy2p.data_ptr(), lens_dev.data_ptr(), dW2.data_ptr(), B, D1, T, ws.data_ptr(), wsb,
                                        _stream()), "ds2_conv2_wgrad_bf16")


def conv2_wgrad_nhwc_bf16(a1n: Tensor, dy2n: Tensor, lens_dev: Tensor, dW2: Tensor):
    """dW2 (32,32,21,11) fp32 from the channels-last bf16 operands a1n (B,D1,T,32), dy2n (B,D2,T,32) (ops.nhwc_bf16 / the fused BatchNorm2d
    kernels' nhwc outputs)."""
    lib = _lib.load()
    B, D1, T, _ = a1n.shape
    assert a1n.dtype == torch.bfloat16 and dy2n.dtype == torch.bfloat16 and a1n.is_contiguous() and dy2n.is_contiguous() and dy2n.shape[2] == T
    wsb = lib.ds2_conv2_wgrad_bf16_workspace_bytes(B, D1)
    ws = _ws(wsb, a1n.device)
    _lib.check(lib.ds2_conv2_wgrad_nhwc_bf16(a1n.data_ptr(), dy2n.data_ptr(), lens_dev.data_ptr(), dW2.data_ptr(), B, D1, T, ws.data_ptr(), wsb,
                                             _stream()), "ds2_conv2_wgrad_nhwc_bf16")


# ------------------------------------------------------------------------------------------------
# recurrence
# ------------------------------------------------------------------------------------------------
def rnn_pack(gates: int, whh: Tensor, bf16=False):
    """W_hh (2, G*H, H) fp32 -> (wp_fwd, wp_bwd) in MFMA-fragment order (fp32 or bf16 fragments; see csrc/rnn.hip).  bf16 = 2: the fp32
    mode with the SPLIT forward recurrence (rnn_fwd(bf16=2)): wp_fwd = [fp32 fragments | bf16 hi fragments | bf16 lo fragments], wp_bwd fp32."""
    _chk_f32(whh)
    assert whh.is_contiguous() and whh.dim() == 3
    lib = _lib.load()
    H = whh.size(2)
    wpf = torch.empty(lib.ds2_rnn_packed_bytes(gates, H, 0, int(bf16)), dtype=torch.uint8, device=whh.device)
    wpb = torch.empty(lib.ds2_rnn_packed_bytes(gates, H, 1, int(bf16)), dtype=torch.uint8, device=whh.device)
    _lib.check(lib.ds2_rnn_pack_whh(gates, whh.data_ptr(), wpf.data_ptr(), wpb.data_ptr(), H, int(bf16), _stream()), "ds2_rnn_pack_whh")
    return wpf, wpb


def rnn_ws(kind: str, gates: int, B: int, H: int, bf16, device) -> Tensor:
    """The workspace of ONE rnn_fwd ("fwd") / rnn_bwd / rnn_bwd_bn ("bwd") call, filled with 0xff bytes NOW, in stream order: pass it as `ws=` to
    the call.  A persistent launch needs its exchange buffers armed with that pattern; armed here — ahead of the GEMM in front of the recurrence —
    the fill is no longer a launch of its own between that GEMM and the launch that waits for every CU (ds2_rnn_ctx.ws_prearmed)."""
    lib = _lib.load()
    n = lib.ds2_rnn_fwd_workspace_bytes(B, H, int(bf16)) if kind == "fwd" else lib.ds2_rnn_bwd_workspace_bytes(gates, B, H, int(bf16))
    ws = _ws(n, device)
    _lib.check(lib.ds2_memset_async(ws.data_ptr(), 0xff, ws.numel(), _stream()), "ds2_memset_async")
    return ws


def _take_ws(ws: Optional[Tensor], need: int, device):
    """(workspace, armed): `ws` from rnn_ws (armed: every byte 0xff) or a fresh, un-armed one"""
    if ws is None:
        return _ws(need, device), False
    assert ws.numel() >= need and ws.device == torch.device(device)
    return ws, True


def _rnn_call(device, armed: bool, thunk):
    """Run one recurrence entry point with the context's one-shot `ws_prearmed` promise set iff the workspace was armed by rnn_ws — and never
    left standing: the library clears it when it reads it; an argument check that fails before that must not hand it to the NEXT call."""
    ctx = rnn_ctx(device)
    ctx.ws_prearmed = 1 if armed else 0
    try:
        return thunk()
    finally:
        ctx.ws_prearmed = 0


def rnn_persistent_enable(forward: bool = True, backward: bool = True, device=None) -> None:
    """Which bf16 recurrences may run as one persistent launch.  A persistent launch needs all of its workgroups resident at once, so
    the backward one must be off while collectives run on a communication stream during backward (data-parallel training)."""
    _lib.check(_lib.load().ds2_rnn_persistent_enable(_ctxp(device), int(bool(forward)), int(bool(backward))), "ds2_rnn_persistent_enable")


def rnn_persistent_check(device=None) -> None:
    """Raise if a persistent recurrence launch starved since the last call (a workgroup never saw its operand, or the launch never became
    resident: it needs every workgroup on the chip at once).  Call at a point where the device is idle anyway (the train step's loss sync)."""
    rec = (C.c_int * 8)()
    _lib.check(_lib.load().ds2_rnn_persistent_status(_ctxp(device), C.cast(rec, C.c_void_p)), "ds2_rnn_persistent_status")
    if rec[0]:
        starved, left = rnn_persistent_counters(device)
        what = {1: "forward", 2: "backward", 3: "census (launch never resident)"}.get(rec[0], str(rec[0]))
        raise _lib.DS2LibraryError(
            f"persistent recurrence starved ({what}): ids ({rec[1]}, {rec[2]}, {rec[3]}) step {rec[4]} wave {rec[5]} pending chunks {rec[6] & 0xffffffff:08x} "
            f"(L2-local exchange: {rec[7]}); the results of that step are invalid.  The next {left if left >= 0 else 'ALL'} recurrence calls run on the "
            f"one-launch-per-step kernels, then the persistent kernels are armed again (DS2_RNN_REARM_CALLS; DS2_RNN_PERSISTENT=0 selects the step "
            f"kernels from the start).  Starved launches since load: {starved}.")


def rnn_poison_if_starved(buf: Tensor) -> None:
    """In stream order, without a host synchronisation: overwrite `buf` (fp32, contiguous) with NaN if a persistent recurrence launch enqueued
    before this call has recorded starvation.  The record stays for the next rnn_persistent_check()."""
    _chk_f32(buf)
    assert buf.is_contiguous()
    _lib.check(_lib.load().ds2_rnn_poison_if_starved(_ctxp(buf.device), buf.data_ptr(), buf.numel(), _stream()), "ds2_rnn_poison_if_starved")


def rnn_poison_seen(device=None) -> bool:
    """True once a rnn_poison_if_starved kernel has actually overwritten a buffer since the last rnn_persistent_check(): a host memory read
    (no synchronisation).  A caller that sees it should run rnn_persistent_check(), which raises, clears the record and moves the next
    recurrence calls onto the step kernels."""
    return bool(_lib.load().ds2_rnn_poison_seen(_ctxp(device)))


def rnn_persistent_counters(device=None):
    """(launches through the calling thread's context that starved, recurrence calls left on the step kernels before re-arming)."""
    out = (C.c_int * 2)()
    _lib.check(_lib.load().ds2_rnn_persistent_counters(_ctxp(device), C.cast(out, C.c_void_p)), "ds2_rnn_persistent_counters")
    return int(out[0]), int(out[1])


def rnn_fwd(gates: int, gx: Tensor, wp_fwd: Tensor, bhh: Tensor, lens_dev: Tensor, T: int, B: int, H: int, bf16=False,
            packed_gates: bool = False, h_bf16: Optional[Tensor] = None, hsum: Optional[Tensor] = None, ws: Optional[Tensor] = None):
    """gx (T*B, 2*G*H) in/out; returns (hbuf (T*B, 2H), aux (T*B, 2H)[, gates_bf (T*B, 2H, 4) bf16]).
    bf16: False / 0 fp32, True / 1 bf16 operands, 2 = fp32 mode with the split persistent kernel (h and W_hh as hi + lo bf16 planes, three MFMAs
    per product: fp32-grade results at the bf16 matrix rate; wp_fwd from rnn_pack(bf16=2); rnn_last_path() & 32 when it took the call).
    packed_gates: the saved-for-backward gates go to one 8-byte bf16 record per hidden unit (gx keeps the x-projections; GRU aux
    is not written) — pass the returned buffer to rnn_bwd.
    h_bf16: optional (T*B, 2H) bf16 buffer that receives a bf16 copy of hbuf — by a persistent launch only (rnn_last_path() & 1).
    hsum: optional (2, ceil(B/16), H) fp32 buffer that receives, per direction and 16-row batch tile, the sums of h over time — by a persistent
    launch only (rnn_last_path() & 1); bf16 training mode (bf16 = 1, packed_gates)."""
    _chk_f32(bhh, hsum)
    assert gx.is_contiguous() and bhh.is_contiguous()
    if h_bf16 is not None:
        assert h_bf16.dtype == torch.bfloat16 and h_bf16.is_contiguous() and h_bf16.numel() == T * B * 2 * H
    lib = _lib.load()
    hbuf = torch.empty(T * B, 2 * H, dtype=torch.float32, device=gx.device)
    aux = torch.empty_like(hbuf)
    rec = torch.empty(T * B, 2 * H, 4, dtype=torch.bfloat16, device=gx.device) if packed_gates else None
    wsb = lib.ds2_rnn_fwd_workspace_bytes(B, H, int(bf16))
    ws, armed = _take_ws(ws, wsb, gx.device)
    if gx.dtype == torch.bfloat16 or hsum is not None:
        # the bf16 training mode's entry point: bf16 x-projections (gemm_bf16_nt_obf16) and / or the per-tile column sums of h.  Persistent
        # launches take bf16 x-projections as they are; anything else (a cooldown after a starved launch, a shape without a persistent
        # kernel) gets them widened and runs as before
        assert int(bf16) == 1 and packed_gates, "rnn_fwd: bf16 x-projections / hsum belong to the bf16 training mode with packed gate records"
        gxb = gx if gx.dtype == torch.bfloat16 else None
        gxf = None if gxb is not None else gx
        for _attempt in (0, 1):
            rc = _rnn_call(gx.device, armed and _attempt == 0, lambda: lib.ds2_rnn_fwd_x(
                _ctxp(gx.device), gates, _ptr(gxf), _ptr(gxb), wp_fwd.data_ptr(), bhh.data_ptr(), hbuf.data_ptr(), aux.data_ptr(),
                lens_dev.data_ptr(), T, B, H, _ptr(rec), _ptr(h_bf16), _ptr(hsum), ws.data_ptr(), wsb, _stream()))
            if rc != 1:
                break
            gxf, gxb = widen_bf16(gxb), None
        _lib.check(rc, "ds2_rnn_fwd_x")
        return hbuf, aux, rec
    _chk_f32(gx)
    _lib.check(_rnn_call(gx.device, armed, lambda: lib.ds2_rnn_fwd_ex(
        _ctxp(gx.device), gates, gx.data_ptr(), wp_fwd.data_ptr(), bhh.data_ptr(), hbuf.data_ptr(), aux.data_ptr(),
        lens_dev.data_ptr(), T, B, H, int(bf16), _ptr(rec), _ptr(h_bf16), ws.data_ptr(), wsb, _stream())), "ds2_rnn_fwd")
    return (hbuf, aux, rec) if packed_gates else (hbuf, aux)


_CORESIDENT = {}


def wgrad_fits_beside_bwd_recurrence(gates: int, H: int) -> bool:
    """Does a workgroup of the co-resident weight-gradient kernel (gemm_bf16_tn_group: one wave of <= 128 registers per SIMD, 128 KB of LDS)
    fit on a CU that already holds a workgroup of the K-split backward recurrence of this shape (two waves per SIMD)?  Decided from the
    register count of the LOADED kernel (512 registers per SIMD lane, allocated in blocks of 8; 160 KB of LDS per CU)."""
    key = (gates, H)
    if key not in _CORESIDENT:
        out = (C.c_int * 3)()
        has = _lib.load().ds2_rnn_bwd_ksplit_footprint(gates, H, out)
        regs = (out[0] + 7) // 8 * 8
        _CORESIDENT[key] = bool(has == 1 and out[2] == 512 and 2 * regs + 128 <= 512 and out[1] + 128 * 1024 <= 160 * 1024)
    return _CORESIDENT[key]


def rnn_last_path(device=None) -> int:
    """bit 0 / bit 1: the last rnn_fwd / rnn_bwd call ran as one persistent launch (and produced its optional outputs); bit 2: that
    backward launch was the K-split kernel (bf16 partial-dh exchange; results within a stated tolerance of the step kernels')"""
    return _lib.load().ds2_rnn_last_path(_ctxp(device))


def rnn_bwd(gates: int, dy: Tensor, gx: Optional[Tensor], aux: Tensor, hbuf: Tensor, wp_bwd: Tensor, lens_dev: Tensor, T: int, B: int, H: int,
            bf16: bool = False, dgx_bf16: Optional[Tensor] = None, gates_bf16: Optional[Tensor] = None, dhn_bf16: Optional[Tensor] = None,
            bias_part: Optional[Tensor] = None, ws: Optional[Tensor] = None):
    """dgx_bf16: optional (T*B, 2*G*H) bf16 buffer that receives dGx (then `gx` keeps the gates).
    gates_bf16: the packed records of rnn_fwd(packed_gates=True), read instead of gx / GRU aux (gx may then be None).
    dhn_bf16 (GRU, (T*B, 2H) bf16 copy of d(hn)) and bias_part ((B, 2, 4, H) fp32 per-batch-row sums over time of the gate gradients):
    optional outputs of a persistent launch only (rnn_last_path() & 2)."""
    _chk_f32(dy, gx, aux, hbuf, bias_part)
    if dgx_bf16 is not None:
        assert dgx_bf16.dtype == torch.bfloat16 and dgx_bf16.is_contiguous() and dgx_bf16.numel() == T * B * 2 * gates * H
    if gates_bf16 is not None:
        assert gates_bf16.dtype == torch.bfloat16 and gates_bf16.is_contiguous() and gates_bf16.numel() == T * B * 2 * H * 4
    if dhn_bf16 is not None:
        assert dhn_bf16.dtype == torch.bfloat16 and dhn_bf16.is_contiguous() and dhn_bf16.numel() == T * B * 2 * H
    if bias_part is not None:
        assert bias_part.is_contiguous() and bias_part.numel() == B * 2 * 4 * H
    assert gx is not None or (dgx_bf16 is not None and gates_bf16 is not None)
    lib = _lib.load()
    wsb = lib.ds2_rnn_bwd_workspace_bytes(gates, B, H, int(bf16))
    ws, armed = _take_ws(ws, wsb, dy.device)
    _lib.check(_rnn_call(dy.device, armed, lambda: lib.ds2_rnn_bwd_ex(
        _ctxp(dy.device), gates, dy.data_ptr(), _row_pitch(dy), _ptr(gx), aux.data_ptr(), hbuf.data_ptr(), wp_bwd.data_ptr(),
        lens_dev.data_ptr(), T, B, H, int(bf16), _ptr(dgx_bf16), _ptr(gates_bf16), _ptr(dhn_bf16), _ptr(bias_part),
        ws.data_ptr(), wsb, _stream())), "ds2_rnn_bwd")


def bn1d_bwd_sums(dY: Tensor, X: Tensor, mean: Tensor, var: Tensor, gamma: Tensor, out: Optional[Tuple[Tensor, Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """Column sums of BatchNorm1d backward only: (sum dY, sum dY * xhat) = (dbeta, dgamma), two (H,) fp32 vectors — written into `out` when
    given (e.g. the gradient buffers themselves).  The elementwise half is left to rnn_bwd_bn."""
    _chk_f32(dY, X, mean, var, gamma)
    lib = _lib.load()
    M, H = X.shape
    if out is None:
        both = torch.empty(2, H, dtype=torch.float32, device=X.device)
        out = (both[0], both[1])
    s0, s1 = out
    _chk_f32(s0, s1)
    assert s0.is_contiguous() and s1.is_contiguous() and s0.numel() == H == s1.numel()
    wsb = lib.ds2_colreduce_workspace_bytes(M, H)
    ws = _ws(wsb, X.device)
    _lib.check(lib.ds2_bn1d_bwd_f32(dY.data_ptr(), _row_pitch(dY), X.data_ptr(), _row_pitch(X), None, 0, M, H, mean.data_ptr(), var.data_ptr(),
                                    gamma.data_ptr(), BN_EPS, s1.data_ptr(), s0.data_ptr(), ws.data_ptr(), wsb, _stream()), "ds2_bn1d_bwd_f32")
    return s0, s1


def rnn_bwd_bn(gates: int, dyn: Tensor, bn_x: Tensor, mean: Tensor, var: Tensor, gamma: Tensor, sums: Tensor, gx: Optional[Tensor], aux: Tensor,
               hbuf: Tensor, wp_bwd: Tensor, lens_dev: Tensor, T: int, B: int, H: int, bf16: bool = False, dgx_bf16: Optional[Tensor] = None,
               gates_bf16: Optional[Tensor] = None, dhn_bf16: Optional[Tensor] = None, bias_part: Optional[Tensor] = None, ws: Optional[Tensor] = None):
    """rnn_bwd for a layer whose output feeds a BatchNorm1d: `dyn` is the gradient wrt the BatchNorm's OUTPUT, `bn_x` its input, `sums` the
    (s0, s1) result of bn1d_bwd_sums.  The K-split kernel applies the elementwise BatchNorm backward on the fly (rnn_last_path() & 16);
    any other kernel family gets it materialised first."""
    s0, s1 = sums
    xbf = bn_x.dtype == torch.bfloat16                  # the centred bf16 operand of center_colstats (mean = its delta): ds2_rnn_bwd_bn_xbf16
    _chk_f32(dyn, None if xbf else bn_x, mean, var, gamma, s0, s1, gx, aux, hbuf, bias_part)
    assert s0.is_contiguous() and s1.is_contiguous() and s0.numel() == H == s1.numel() and bn_x.shape[0] == T * B and bn_x.shape[1] >= H and dyn.shape == (T * B, H)
    assert bn_x.stride(1) == 1 and (not xbf or bn_x.shape[1] == H or bn_x.stride(0) % 2 == 0)
    lib = _lib.load()
    wsb = lib.ds2_rnn_bwd_workspace_bytes(gates, B, H, int(bf16))
    ws, armed = _take_ws(ws, wsb, dyn.device)
    def call(scratch):
        return _rnn_call(dyn.device, armed and scratch is None, lambda: (lib.ds2_rnn_bwd_bn_xbf16 if xbf else lib.ds2_rnn_bwd_bn)(_ctxp(dyn.device), gates, dyn.data_ptr(), _row_pitch(dyn), bn_x.data_ptr(), bn_x.stride(0), mean.data_ptr(), var.data_ptr(),
                                  gamma.data_ptr(), s0.data_ptr(), s1.data_ptr(), BN_EPS, _ptr(scratch), _ptr(gx), aux.data_ptr(),
                                  hbuf.data_ptr(), wp_bwd.data_ptr(), lens_dev.data_ptr(), T, B, H, int(bf16), _ptr(dgx_bf16), _ptr(gates_bf16),
                                  _ptr(dhn_bf16), _ptr(bias_part), ws.data_ptr(), wsb, _stream()))
    rc = call(None)                  # the fused K-split launch needs no scratch: the (T*B, H) fp32 buffer is allocated only on the fallback (rc 1)
    if rc == 1:
        rc = call(torch.empty(T * B, H, dtype=torch.float32, device=dyn.device))
    _lib.check(rc, "ds2_rnn_bwd_bn")


def rnn_bias_grads(gates: int, bias_part: Tensor, dbih: Tensor, dbhh: Tensor):
    """bias_part (B, 2, 4, H) of rnn_bwd -> dbih (2*G*H,) / dbhh (2, G*H) gradient buffers (contiguous)."""
    _chk_f32(bias_part, dbih, dbhh)
    B, _, _, H = bias_part.shape
    assert bias_part.is_contiguous() and dbih.is_contiguous() and dbhh.is_contiguous() and dbih.numel() == 2 * gates * H == dbhh.numel()
    _lib.check(_lib.load().ds2_rnn_bias_grads(gates, bias_part.data_ptr(), B, H, dbih.data_ptr(), dbhh.data_ptr(), _stream()), "ds2_rnn_bias_grads")


# ------------------------------------------------------------------------------------------------
# CTC
# ------------------------------------------------------------------------------------------------
def ctc_loss(logits: Tensor, targets_dev: Tensor, tgt_off_dev: Tensor, in_lens_dev: Tensor, tgt_lens_dev: Tensor, max_tgt: int,
             grad_scale: float, want_grad: bool = True):
    """logits (T,B,C) (last dim contiguous, uniform row pitch).  Returns (nll (B,), grad (T,B,C) or None)."""
    _chk_f32(logits)
    lib = _lib.load()
    T, B, Cc = logits.shape
    assert logits.stride(2) == 1 and logits.stride(0) == B * logits.stride(1)
    ld = logits.stride(1)
    nll = torch.empty(B, dtype=torch.float32, device=logits.device)
    grad = torch.empty(T, B, Cc, dtype=torch.float32, device=logits.device) if want_grad else None
    wsb = lib.ds2_ctc_workspace_bytes(T, B, max_tgt)
    ws = _ws(wsb, logits.device)
    _lib.check(lib.ds2_ctc_loss_f32(logits.data_ptr(), ld, T, B, Cc, targets_dev.data_ptr(), tgt_off_dev.data_ptr(),
                                    in_lens_dev.data_ptr(), tgt_lens_dev.data_ptr(), int(max_tgt), nll.data_ptr(), _ptr(grad), Cc,
                                    float(grad_scale), ws.data_ptr(), wsb, _stream()), "ds2_ctc_loss_f32")
    return nll, grad


def ctc_batch_mean(nll: Tensor) -> Tensor:
    """(1,) fp32 = nll.sum() / B, on the device in a fixed order (trainers/deepspeech_trainer.py:110-112)."""
    _chk_f32(nll)
    assert nll.dim() == 1 and nll.is_contiguous()
    out = torch.empty(1, dtype=torch.float32, device=nll.device)
    _lib.check(_lib.load().ds2_ctc_batch_mean_f32(nll.data_ptr(), nll.numel(), out.data_ptr(), _stream()), "ds2_ctc_batch_mean_f32")
    return out


def add_i64(x: Tensor, v: int = 1):
    """x += v for a small contiguous int64 vector (the BatchNorm num_batches_tracked counters)."""
    assert x.dtype == torch.int64 and x.is_contiguous() and x.is_cuda
    _lib.check(_lib.load().ds2_add_i64(x.data_ptr(), x.numel(), int(v), _stream()), "ds2_add_i64")


def softmax_rows(x: Tensor) -> Tensor:
    """softmax over the last dim of a (rows, C) row-major view."""
    _chk_f32(x)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().ds2_softmax_rows_f32(x.data_ptr(), _row_pitch(x), y.data_ptr(), x.size(1), x.size(0), x.size(1), _stream()),
               "ds2_softmax_rows_f32")
    return y


def greedy_decode(probs: Tensor, sizes: Tensor | None = None, blank: int = 0):
    """probs (B,T,C) fp32 on the GPU -> (ids (B,T) i32, offsets (B,T) i32, lengths (B) i32), all on the GPU."""
    _chk_f32(probs)
    if probs.dim() != 3 or probs.stride(2) != 1:
        raise ValueError("greedy_decode: probs must be (B,T,C) with a contiguous class dim")
    B, T, C = probs.shape
    dev = probs.device
    if sizes is not None:
        sizes = sizes.to(device=dev, dtype=torch.int32).contiguous()
    ids = torch.empty((B, T), dtype=torch.int32, device=dev)
    offs = torch.empty((B, T), dtype=torch.int32, device=dev)
    lens = torch.empty((B,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    n = lib.ds2_greedy_decode_workspace_bytes(B, T)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.check(lib.ds2_greedy_decode_f32(probs.data_ptr(), probs.stride(0), probs.stride(1), B, T, C,
                                         sizes.data_ptr() if sizes is not None else None, blank, ids.data_ptr(), offs.data_ptr(),
                                         lens.data_ptr(), ws.data_ptr(), n, _stream()), "ds2_greedy_decode_f32")
    return ids, offs, lens


_BASIS_CACHE = {}


def dft_basis(n_fft: int, window: str, device) -> Tensor:
    """(n_fft, 2*(n_fft/2+1)) fp32: column 2j = w[k] cos(2 pi k j / n_fft), column 2j+1 = -w[k] sin(...); w = scipy's periodic window
    (what librosa.filters.get_window returns for a name), computed in float64 once per (n_fft, window, device)."""
    key = (n_fft, window, str(device))
    if key not in _BASIS_CACHE:
        import numpy as np
        from scipy.signal import get_window
        w = get_window(window, n_fft, fftbins=True).astype(np.float64)
        k = np.arange(n_fft)[:, None]
        j = np.arange(n_fft // 2 + 1)[None, :]
        ang = 2.0 * np.pi * ((k * j) % n_fft) / n_fft
        basis = np.empty((n_fft, 2 * (n_fft // 2 + 1)))
        basis[:, 0::2] = w[:, None] * np.cos(ang)
        basis[:, 1::2] = -w[:, None] * np.sin(ang)
        _BASIS_CACHE[key] = torch.from_numpy(basis.astype(np.float32)).to(device)
    return _BASIS_CACHE[key]


def spectrogram(audio: Tensor, n_samples: Tensor, n_fft: int, hop: int, window: str = "hamming", pad_mode: str = "constant",
                normalize: bool = False, frames: int = 0):
    """audio (B, L) fp32 GPU waveforms (rows zero/garbage beyond n_samples) -> (spect (B,1,n_fft/2+1,T), frames (B,) int32 CPU)."""
    _chk_f32(audio)
    assert audio.dim() == 2 and audio.stride(1) == 1
    lib = _lib.load()
    B = audio.size(0)
    n_host = [int(v) for v in n_samples.tolist()]
    fr = [lib.ds2_spectrogram_frames(n, hop) for n in n_host]
    T = frames or max(max(fr), 1)
    n_dev = torch.as_tensor(n_host, dtype=torch.int32).to(audio.device)
    out = torch.empty((B, 1, n_fft // 2 + 1, T), dtype=torch.float32, device=audio.device)
    wsb = lib.ds2_spectrogram_workspace_bytes(B, T, n_fft, hop)
    ws = _ws(wsb, audio.device)
    _lib.check(lib.ds2_spectrogram_f32(audio.data_ptr(), audio.stride(0), n_dev.data_ptr(), B, T, n_fft, hop,
                                       dft_basis(n_fft, window, audio.device).data_ptr(), {"constant": 0, "reflect": 1}[pad_mode],
                                       int(bool(normalize)), out.data_ptr(), ws.data_ptr(), wsb, _stream()), "ds2_spectrogram_f32")
    return out, torch.tensor([min(f, T) for f in fr], dtype=torch.int32)


# ------------------------------------------------------------------------------------------------
# optimizer
# ------------------------------------------------------------------------------------------------
def adamw(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
          weight_decay: float = 1e-5, grad_scale: float = 1.0, apply_flag: Optional[Tensor] = None):
    """apply_flag: optional int32 GPU tensor read by the kernel when it runs (0 = leave p, m, v untouched): step_gate()."""
    _chk_f32(p, g, m, v)
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    assert apply_flag is None or (apply_flag.dtype == torch.int32 and apply_flag.is_cuda)
    _lib.check(_lib.load().ds2_adamw_gated_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, betas[0], betas[1],
                                               eps, weight_decay, int(step), grad_scale, _ptr(apply_flag), _stream()), "ds2_adamw_gated_f32")


def step_gate(loss: Tensor) -> Tensor:
    """int32 GPU tensor [1]: 1 if, when the stream gets there, `loss` (0-d / 1-element fp32 GPU tensor) is finite and non-negative and no
    persistent recurrence launch has recorded starvation — the device-side form of check_loss + rnn_persistent_check."""
    _chk_f32(loss)
    flag = torch.empty(1, dtype=torch.int32, device=loss.device)
    _lib.check(_lib.load().ds2_rnn_step_gate(_ctxp(loss.device), loss.data_ptr(), flag.data_ptr(), _stream()), "ds2_rnn_step_gate")
    return flag
