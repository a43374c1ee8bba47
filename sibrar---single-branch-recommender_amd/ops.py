"""torch.autograd wrappers around the C-ABI kernels (include/sibrar_hip.h).

PyTorch is used for device memory, streams and the autograd tape only; every forward/backward computation is a HIP
kernel of libsibrar_hip.so. All tensors are fp32 / contiguous CUDA tensors unless stated otherwise. There is no CPU path:
calling any op with CPU tensors raises.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from ._lib import call, lib, ptr, stream

ACT_CODES = {None: 0, 'none': 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3, 'selu': 4}
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
NORM_EPS = 1e-12
PARTITION_MAX_MODALITIES = 8      # sbr_partition_slots (SBR_PART_MAX in rowops.hip)
COLRED_WS_FACTOR = 17             # column-reduction workspaces: totals + SBR_COLRED_REP replicas (common.h)


def act_code(act) -> int:
    if isinstance(act, torch.nn.Module):
        act = act.__class__.__name__.lower()
    if act not in ACT_CODES:
        raise ValueError(f'unsupported activation {act!r}')
    return ACT_CODES[act]


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('sibrar HIP ops need CUDA(HIP) tensors; there is no CPU fallback in this package')


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class KernelTimer:
    """Optional HIP-event timing of individual kernel launches on the stream they are launched on (torch's current stream is
    the stream handed to the C ABI). bench.py enables it over plain-launch steps to measure every kernel of the step live:
    ``('call', entry point)`` for every C-ABI call, and the GEMM / scorer wrappers add a key that carries the shape."""
    enabled = False
    records = {}          # key -> [(start_event, stop_event)]

    @classmethod
    def reset(cls, enabled: bool):
        from . import _lib
        cls.enabled, cls.records = enabled, {}
        _lib.CALL_TIMER = cls._bracket if enabled else None

    @classmethod
    def _bracket(cls, name, thunk):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = thunk()
        b.record()
        cls.records.setdefault(('call', name), []).append((a, b))
        key = getattr(_TIMED_KEY, 'value', None)
        if key is not None:              # the shape-carrying key of the wrapper that made this call: the same event pair
            cls.records.setdefault(key, []).append((a, b))
        return rc

    @classmethod
    def results(cls):
        torch.cuda.synchronize()
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in cls.records.items()}


import threading as _threading

_TIMED_KEY = _threading.local()


def _timed(key, fn):
    """``fn`` makes ONE C-ABI call; while KernelTimer is on, the events around that call are also filed under ``key``."""
    if not KernelTimer.enabled:
        return fn()
    _TIMED_KEY.value = key
    try:
        return fn()
    finally:
        _TIMED_KEY.value = None


# ---- raw kernel helpers (no autograd) -------------------------------------------------------------------------------
def gemm(mode: int, A, lda, a_idx, B, ldb, b_idx, bias, C, ldc, c_idx, M, N, K, act=0, atomic=0):
    _timed(('gemm_f32', mode, M, N, K, a_idx is not None or b_idx is not None),
           lambda: call('sbr_gemm_f32', mode, ptr(A), lda, ptr(a_idx), ptr(B), ldb, ptr(b_idx), ptr(bias), ptr(C), ldc,
                        ptr(c_idx), M, N, K, act, atomic, stream()))


import os as _os

_WRES = True                    # weights-resident kernel for N = K = 128 products below the bf16-split kernel's row threshold (tests switch it)
# fp32 products on the bf16 matrix pipe over exact three-way operand splits (csrc/gemm_split_f32.hip); '0' keeps every product on
# the fp32 pipe (v_mfma_f32_32x32x2_f32)
_SPLIT = _os.environ.get('SBR_GEMM_SPLIT', '1') != '0'
_SPLIT_MIN_ROWS = 4096          # below this the per-workgroup weight set-up is not amortised
_WIDE_HEURISTIC = True          # tests switch it off to reach the wide kernel at small shapes


def _mlp_kernel(M, N, K) -> str:
    """Entry point for an N = K = 128 product without gathers (checked by ``_wres_ok``)."""
    if _SPLIT and M >= _SPLIT_MIN_ROWS and lib().sbr_gemm_split_supported(int(M), int(N), int(K)):
        return 'sbr_gemm_split_f32'
    return 'sbr_gemm_wres_f32'


def _wres_ok(M, N, K, *tensors) -> bool:
    """The weights-resident kernel (csrc/gemm_wres_f32.hip) takes the shared MLP's own products: N = K = 128, no gathers,
    16-byte aligned rows."""
    if not _WRES or M < 1 or not lib().sbr_gemm_wres_supported(int(M), int(N), int(K)):
        return False
    return all(t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and t.stride(1) == 1 for t in tensors)


def wide_pays(M, N, K) -> bool:
    """Shapes at which the wide bf16-split kernel is used when it could be (measured against the fp32 ring kernel,
    tools/lab/wide_time.py): enough 256 x 256 tiles to fill the chip's 256 workgroup slots, and a wide or deep product (256 x 256
    layers run as fast on the fp32 pipe). bench.py prices a GEMM signature with the same rule."""
    return not _WIDE_HEURISTIC or (-(-M // 256) * (N // 256) >= 176 and max(N, K) >= 512)


def _wide_ok(M, N, K, a, w, out, nt: bool) -> bool:
    """The wide bf16-split kernel (csrc/gemm_split_wide_f32.hip) takes the product: enough rows, N = 256 i, K = 32 j >= 64, 16-byte
    aligned rows of A (and of W for NT), unit column strides."""
    if not _SPLIT or M < _SPLIT_MIN_ROWS or not lib().sbr_gemm_split_wide_supported(int(M), int(N), int(K)):
        return False
    if not wide_pays(M, N, K):
        return False
    ok = a.data_ptr() % 16 == 0 and a.stride(0) % 4 == 0 and a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    return ok and (not nt or (w.data_ptr() % 16 == 0 and w.stride(0) % 4 == 0))


def linear_nt_stats_ok(x, W, out) -> bool:
    """True when ``linear_nt(x, W, ..., out=out, stats_ws=ws)`` can leave the batch statistics of its output pending in ``ws``
    (the bf16-split kernel takes the product: N = K = 128, no gathers, enough rows)."""
    M, (N, K) = x.shape[0], W.shape
    return _wres_ok(M, N, K, x, W, out) and _mlp_kernel(M, N, K) == 'sbr_gemm_split_f32'


def linear_nt(x, W, bias=None, act=0, a_idx=None, out=None, c_idx=None, n_rows=None, stats_ws=None, bn_fin=None):
    """out[ci(m)] = act(x[ai(m)] @ W^T + bias). W: [N, K] with arbitrary row stride (column-major weights are handled by
    the caller through csr kernels, not here). ``stats_ws`` (check ``linear_nt_stats_ok`` first): a zeroed column-reduction
    workspace of 17 * 2 * N doubles that receives the per-column sums and sums of squares of ``out`` (``bn_finalize_stats``).
    ``bn_fin`` (with ``stats_ws``) = (arrive, running_mean, running_var, num_batches_tracked, save_mean, save_rstd, eps, momentum):
    the same launch also finalises the BatchNorm statistics (``arrive``: a zeroed int64[1] owned by the BatchNorm)."""
    M = n_rows if n_rows is not None else (a_idx.numel() if a_idx is not None else x.shape[0])
    N, K = W.shape
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    if a_idx is None and c_idx is None and _wres_ok(M, N, K, x, W, out):
        kern = _mlp_kernel(M, N, K)
        if stats_ws is not None and kern != 'sbr_gemm_split_f32':
            raise ValueError('linear_nt(stats_ws=...): this product does not take the kernel with the statistics epilogue')
        if bn_fin is not None:
            if stats_ws is None:
                raise ValueError('linear_nt(bn_fin=...) needs stats_ws')
            arrive, rm, rv, nbt, mean, rstd, eps, mom = bn_fin
            _timed(('gemm_f32', 0, M, N, K, False),
                   lambda: call('sbr_gemm_split_bnstats_f32', ptr(x), x.stride(0), ptr(W), W.stride(0), ptr(bias), ptr(out), out.stride(0),
                                M, N, K, act, ptr(stats_ws), ptr(arrive), ptr(rm), ptr(rv), ptr(nbt), ptr(mean), ptr(rstd), eps, mom,
                                stream()))
            return out
        _timed(('gemm_f32', 0, M, N, K, False),
               lambda: call(kern, 0, ptr(x), x.stride(0), ptr(W), W.stride(0), ptr(bias), ptr(out), out.stride(0), M, N, K,
                            act, None, 0, ptr(stats_ws), stream()))
        return out
    if stats_ws is not None:
        raise ValueError('linear_nt(stats_ws=...): this product does not take the kernel with the statistics epilogue')
    if (_SPLIT and M >= _SPLIT_MIN_ROWS and lib().sbr_gemm_split_proj_supported(int(M), int(N), int(K))
            and all(t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and t.stride(1) == 1 for t in (x, W)) and out.stride(1) == 1):
        # dense modality projector (N = 128, K = 768 / 1024 / 2048 ...) on the bf16 matrix pipe, gather and scatter fused
        _timed(('gemm_f32', 0, M, N, K, a_idx is not None),
               lambda: call('sbr_gemm_split_proj_f32', ptr(x), x.stride(0), ptr(a_idx), ptr(W), W.stride(0), ptr(bias), ptr(out),
                            out.stride(0), ptr(c_idx), M, N, K, act, stream()))
        return out
    if _wide_ok(M, N, K, x, W, out, nt=True):
        # wide layers (N = 256 i: hidden widths 256 / 512, the projectors of C = 256 / 512) on the bf16 matrix pipe, gather and scatter fused
        _timed(('gemm_f32', 0, M, N, K, a_idx is not None),
               lambda: call('sbr_gemm_split_wide_f32', 0, ptr(x), x.stride(0), ptr(a_idx), ptr(W), W.stride(0), ptr(bias), ptr(out),
                            out.stride(0), ptr(c_idx), M, N, K, act, stream()))
        return out
    ws_bytes = lib().sbr_gemm_nt_splitk_workspace(M, N, K) if M > 0 else 0
    if ws_bytes > 0:
        # few output tiles, long K (the modality projectors at small batches): K split over workgroups, deterministic reduce
        ws = _tn_workspace(x.device, ws_bytes)
        _timed(('gemm_f32', 0, M, N, K, a_idx is not None),
               lambda: call('sbr_gemm_nt_splitk_f32', ptr(x), x.stride(0), ptr(a_idx), ptr(W), W.stride(0), ptr(bias), ptr(out),
                            out.stride(0), ptr(c_idx), M, N, K, act, ptr(ws), ws.numel() * 4, stream()))
        return out
    gemm(0, x, x.stride(0), a_idx, W, W.stride(0), None, bias, out, out.stride(0), c_idx, M, N, K, act, 0)
    return out


def matmul_nn(dz, W, a_idx=None, n_rows=None, out=None):
    """dz[ai(m)] @ W, W: [K, N] row-major."""
    M = n_rows if n_rows is not None else (a_idx.numel() if a_idx is not None else dz.shape[0])
    K, N = W.shape
    if out is None:
        out = torch.empty(M, N, device=dz.device, dtype=torch.float32)
    if a_idx is None and _wres_ok(M, N, K, dz, W, out):
        _timed(('gemm_f32', 1, M, N, K, False),
               lambda: call(_mlp_kernel(M, N, K), 1, ptr(dz), dz.stride(0), ptr(W), W.stride(0), None, ptr(out), out.stride(0), M, N, K, 0,
                            None, 0, None, stream()))
        return out
    if _wide_ok(M, N, K, dz, W, out, nt=False):
        _timed(('gemm_f32', 1, M, N, K, a_idx is not None),
               lambda: call('sbr_gemm_split_wide_f32', 1, ptr(dz), dz.stride(0), ptr(a_idx), ptr(W), W.stride(0), None, ptr(out),
                            out.stride(0), None, M, N, K, 0, stream()))
        return out
    gemm(1, dz, dz.stride(0), a_idx, W, W.stride(0), None, None, out, out.stride(0), None, M, N, K, 0, 0)
    return out


def matmul_nn_actgrad_ok(dz, W, y, out) -> bool:
    K, N = W.shape
    return _wres_ok(dz.shape[0], N, K, dz, W, y, out)


def matmul_nn_actgrad(dz, W, y, act: int, out, colsum_ws=None):
    """out = (dz @ W) * act'(y) — the gradient at the pre-activation of the layer in front (whose OUTPUT is y) in one kernel;
    ``colsum_ws``: that layer's bias gradient is left pending there (``colred_finish``). Check ``matmul_nn_actgrad_ok`` first."""
    M = dz.shape[0]
    K, N = W.shape
    _timed(('gemm_f32', 1, M, N, K, False),
           lambda: call(_mlp_kernel(M, N, K), 1, ptr(dz), dz.stride(0), ptr(W), W.stride(0), None, ptr(out), out.stride(0), M, N, K, act,
                        ptr(y), y.stride(0), ptr(colsum_ws), stream()))
    return out


_TN_WS = {}
# Workspaces that were outgrown. Their device addresses are baked into every hipGraph captured while they were current
# (engine.FusedTrainStep), so they are never handed back to the caching allocator: a captured step keeps writing its split-K
# slabs into the block it was captured with, and nothing else can be placed there. Growth is geometric, so the retired blocks
# together are smaller than the current one.
_WS_RETIRED = []
WS_GENERATION = 0                 # bumped on every workspace reallocation (tests; diagnostics)


def _grow(table, key, need, make):
    global WS_GENERATION
    old = table.get(key)
    if old is not None:
        _WS_RETIRED.append(old)
        need = max(need, 2 * old.numel())
    table[key] = ws = make(need)
    WS_GENERATION += 1
    return ws


def _tn_workspace(device, nbytes):
    """Grow-only scratch for the split-K slabs (one per device; reused by every dW product of a step)."""
    ws = _TN_WS.get(device)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _grow(_TN_WS, device, max((nbytes + 3) // 4, 1 << 20), lambda n: torch.empty(n, device=device, dtype=torch.float32))
    return ws


def matmul_tn(dz, x, a_idx=None, b_idx=None, n_rows=None, out=None):
    """sum_r dz[ai(r)]^T x[bi(r)] -> [dz.shape[1], x.shape[1]] (split-K over r with a deterministic slab reducer)."""
    from ._lib import lib
    R = n_rows if n_rows is not None else dz.shape[0]
    M, N = dz.shape[1], x.shape[1]
    if out is None:
        out = torch.empty(M, N, device=dz.device, dtype=torch.float32)
    if R == 0:
        return out.zero_()
    ws_bytes = lib().sbr_gemm_tn_f32_workspace(M, N, R)
    ws = _tn_workspace(dz.device, ws_bytes)
    _timed(('gemm_f32', 2, M, N, R, a_idx is not None or b_idx is not None),
           lambda: call('sbr_gemm_tn_f32', ptr(dz), dz.stride(0), ptr(a_idx), ptr(x), x.stride(0), ptr(b_idx), ptr(out),
                        out.stride(0), M, N, R, ptr(ws), ws.numel() * 4, stream()))
    return out


class DeferredTN:
    """dW products of one backward pass whose split-K slabs are summed by ONE launch at the end (sbr_splitk_reduce_multi).
    Every product keeps its own persistent slab workspace under a caller-chosen key: captured step graphs hold the addresses, so an
    outgrown workspace is retired (kept allocated), never freed. (Measured and dropped: the slab launches themselves as one grouped
    launch at ``finish()`` — 5 us slower per c2 step than one launch per product)"""

    def __init__(self):
        self.ws = {}                 # key -> float32 workspace tensor
        self.pending = []            # (workspace, out, M, N, splits)

    def matmul_tn(self, key, dz, x, a_idx=None, b_idx=None, n_rows=None, out=None):
        import ctypes
        R = n_rows if n_rows is not None else dz.shape[0]
        M, N = dz.shape[1], x.shape[1]
        if R == 0:
            return out.zero_()
        need = lib().sbr_gemm_tn_f32_workspace(M, N, R)
        ws = self.ws.get(key)
        if ws is None or ws.numel() * 4 < need:
            if ws is not None:
                _WS_RETIRED.append(ws)
            ws = self.ws[key] = torch.empty(max((need + 3) // 4, 2 * ws.numel() if ws is not None else 0), device=dz.device,
                                            dtype=torch.float32)
        splits = ctypes.c_int(0)
        _timed(('gemm_f32', 2, M, N, R, a_idx is not None or b_idx is not None),
               lambda: call('sbr_gemm_tn_f32_slabs', ptr(dz), dz.stride(0), ptr(a_idx), ptr(x), x.stride(0), ptr(b_idx), M, N, R,
                            ptr(ws), ws.numel() * 4, ctypes.cast(ctypes.pointer(splits), ctypes.c_void_p), stream()))
        self.pending.append((ws, out, M, N, splits.value))
        return out

    def finish(self, colred=None):
        """Sums the slabs of the pending products. ``colred`` (optional): [(workspace, out float vector [C])] of folded column sums
        — at most 8, with at most 8 pending products — finished by the same launch (``sbr_splitk_reduce_multi_fin``) instead of a
        ``colred_finish`` launch of their own; returns True when they were taken."""
        import ctypes
        took = False
        fuse = colred and 0 < len(colred) <= 8 and 0 < len(self.pending) <= 8
        for lo in range(0, len(self.pending), 8):
            part = self.pending[lo:lo + 8]
            n = len(part)
            arr = lambda ct, vals, m=n: ctypes.cast((ct * m)(*vals), ctypes.c_void_p)
            args = (n, arr(ctypes.c_void_p, [p[0].data_ptr() for p in part]),
                    arr(ctypes.c_void_p, [p[1].data_ptr() for p in part]), arr(ctypes.c_long, [p[1].stride(0) for p in part]),
                    arr(ctypes.c_int, [p[2] for p in part]), arr(ctypes.c_int, [p[3] for p in part]),
                    arr(ctypes.c_int, [p[4] for p in part]))
            # timed as one launch; the key carries every product's (M, N, slabs) so that a reader can apportion it by slab bytes
            key = ('splitk_reduce_multi', tuple((p[2], p[3], p[4]) for p in part))
            if fuse:
                m = len(colred)
                fin = (m, arr(ctypes.c_void_p, [w.data_ptr() for w, _ in colred], m), arr(ctypes.c_void_p, [o.data_ptr() for _, o in colred], m),
                       arr(ctypes.c_int, [o.numel() for _, o in colred], m))
                _timed(key, lambda: call('sbr_splitk_reduce_multi_fin', *args, *fin, stream()))
                took = True
            else:
                _timed(key, lambda: call('sbr_splitk_reduce_multi', *args, stream()))
        self.pending = []
        return took


_COLSUM_WS = {}


def colsum(x: torch.Tensor, out=None) -> torch.Tensor:
    n, C = x.shape
    if out is None:
        out = torch.empty(C, device=x.device, dtype=torch.float32)
    ws = _COLSUM_WS.get((x.device, C))
    if ws is None:
        ws = _COLSUM_WS[(x.device, C)] = torch.zeros(COLRED_WS_FACTOR * C, device=x.device, dtype=torch.float64)
    call('sbr_colsum', ptr(x), x.stride(0), n, C, ptr(out), ptr(ws), stream())
    return out


def act_grad(dy, y, act: int, idx=None, n_rows=None):
    """dz[j] = dy[idx[j]] * act'(y[idx[j]]) (compact [n, C])."""
    n = n_rows if n_rows is not None else (idx.numel() if idx is not None else dy.shape[0])
    C = dy.shape[1]
    dz = torch.empty(n, C, device=dy.device, dtype=torch.float32)
    call('sbr_act_grad_gather', ptr(dy), ptr(y), dy.stride(0), ptr(idx), ptr(dz), C, n, C, act, stream())
    return dz


def colsum_supported(C: int) -> bool:
    """Column sums folded into the producer of their input (sbr_act_grad_gather_colsum, sbr_bn_score_bwd_apply)."""
    return bool(lib().sbr_act_grad_colsum_supported(int(C)))


def new_colsum_ws(device, C: int) -> torch.Tensor:
    """Zeroed column-reduction workspace (totals + replicas) of one pending reduction; its finishing kernel re-zeroes it."""
    return torch.zeros(COLRED_WS_FACTOR * C, device=device, dtype=torch.float64)


def act_grad_colsum(dy, y, act: int, ws, idx=None, n_rows=None):
    """act_grad that also leaves the column sums of its result pending in ``ws`` (complete them with ``colred_finish``)."""
    n = n_rows if n_rows is not None else (idx.numel() if idx is not None else dy.shape[0])
    C = dy.shape[1]
    dz = torch.empty(n, C, device=dy.device, dtype=torch.float32)
    call('sbr_act_grad_gather_colsum', ptr(dy), ptr(y), dy.stride(0), ptr(idx), ptr(dz), C, n, C, act, ptr(ws), stream())
    return dz


def colred_finish(pending) -> None:
    """pending: [(workspace, out float vector [C])] of folded column sums -> one launch per 8 of them."""
    import ctypes
    for lo in range(0, len(pending), 8):
        part = pending[lo:lo + 8]
        n = len(part)
        wsa = (ctypes.c_void_p * n)(*[w.data_ptr() for w, _ in part])
        outa = (ctypes.c_void_p * n)(*[o.data_ptr() for _, o in part])
        ca = (ctypes.c_int * n)(*[o.numel() for _, o in part])
        call('sbr_colred_finish', n, ctypes.cast(wsa, ctypes.c_void_p), ctypes.cast(outa, ctypes.c_void_p),
             ctypes.cast(ca, ctypes.c_void_p), stream())


# ---- Linear (+ activation) ----------------------------------------------------------------------------------------------
class LinearActFn(Function):
    """y = act(x @ W^T + b) — nn.Linear + activation of modules/polylinear.py:51,63-72."""

    @staticmethod
    def forward(ctx, x, weight, bias, act: int):
        _need_cuda(x, weight)
        x = _f32c(x)
        w = weight if weight.stride(1) == 1 else weight.contiguous()
        y = linear_nt(x, w, bias, act)
        ctx.act = act
        ctx.save_for_backward(x, w, y)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _f32c(dy)
        dz = act_grad(dy, y, ctx.act) if ctx.act else dy
        dx = matmul_nn(dz, w) if ctx.needs_input_grad[0] else None
        dw = matmul_tn(dz, x) if ctx.needs_input_grad[1] else None
        db = colsum(dz) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


# ---- BatchNorm1d (+ activation) ---------------------------------------------------------------------------------------
class BatchNormActFn(Function):
    """Train-mode BatchNorm1d over rows followed by an activation (polylinear.py:61-65; sgd_alg.py:1837)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches_tracked, act: int):
        _need_cuda(x, weight)
        x = _f32c(x)
        n, D = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(D, device=x.device, dtype=torch.float32)
        rstd = torch.empty(D, device=x.device, dtype=torch.float32)
        ws = torch.zeros(COLRED_WS_FACTOR * 2 * D, device=x.device, dtype=torch.float64)
        call('sbr_bn_train_fwd', ptr(x), ptr(y), n, D, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var),
             ptr(num_batches_tracked), ptr(mean), ptr(rstd), ptr(ws), BN_EPS, BN_MOMENTUM, act, stream())
        ctx.act = act
        ctx.save_for_backward(x, y, weight, mean, rstd)
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, num_batches_tracked) if t is not None])
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, mean, rstd = ctx.saved_tensors
        dy = _f32c(dy)
        n, D = x.shape
        dx = torch.empty_like(x)
        dw = torch.empty(D, device=x.device, dtype=torch.float32)
        db = torch.empty(D, device=x.device, dtype=torch.float32)
        ws = torch.zeros(COLRED_WS_FACTOR * 2 * D, device=x.device, dtype=torch.float64)
        call('sbr_bn_train_bwd', ptr(dy), ptr(y), ptr(x), ptr(dx), n, D, ptr(weight), ptr(mean), ptr(rstd), ptr(dw), ptr(db),
             ptr(ws), ctx.act, stream())
        return dx, dw, db, None, None, None, None


def batch_norm_eval(x, weight, bias, running_mean, running_var, act: int):
    _need_cuda(x)
    x = _f32c(x)
    y = torch.empty_like(x)
    call('sbr_bn_eval_fwd', ptr(x), ptr(y), x.shape[0], x.shape[1], ptr(weight), ptr(bias), ptr(running_mean),
         ptr(running_var), BN_EPS, act, stream())
    return y


# ---- row normalisation, dropout, aggregation ---------------------------------------------------------------------------
class L2NormalizeFn(Function):
    """F.normalize(x, p=2, dim=-1) — sgd_alg.py:1873-1874."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = _f32c(x)
        y = torch.empty_like(x)
        inv = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        call('sbr_l2norm_fwd', ptr(x), ptr(y), ptr(inv), x.shape[0], x.shape[1], NORM_EPS, stream())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(y)
        call('sbr_l2norm_bwd', ptr(dy), ptr(y), ptr(inv), ptr(dx), y.shape[0], y.shape[1], NORM_EPS, stream())
        return dx


class DropoutFn(Function):
    """nn.Dropout(p) in training mode with a counter-based mask (seed, element index)."""

    @staticmethod
    def forward(ctx, x, p: float, seed: int):
        _need_cuda(x)
        x = _f32c(x)
        y = torch.empty_like(x)
        call('sbr_dropout', ptr(x), ptr(y), x.numel(), p, seed, stream())
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        dx = torch.empty_like(dy)
        call('sbr_dropout', ptr(dy), ptr(dx), dy.numel(), ctx.p, ctx.seed, stream())
        return dx, None, None


class AggregateFn(Function):
    """mean / max over the k sampled modalities: [S, k, D] -> [S, D] (sgd_alg.py:27-31, 1861)."""

    @staticmethod
    def forward(ctx, e, mode: int):
        _need_cuda(e)
        e = _f32c(e)
        S, k, D = e.shape
        out = torch.empty(S, D, device=e.device, dtype=torch.float32)
        arg = torch.empty(S, D, device=e.device, dtype=torch.uint8) if mode == 1 else None
        call('sbr_aggregate_fwd', ptr(e), ptr(out), ptr(arg), S, k, D, mode, stream())
        ctx.mode, ctx.shape = mode, (S, k, D)
        ctx.save_for_backward(arg if arg is not None else torch.empty(0, device=e.device))
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        S, k, D = ctx.shape
        dout = _f32c(dout)
        de = torch.empty(S, k, D, device=dout.device, dtype=torch.float32)
        call('sbr_aggregate_bwd', ptr(dout), ptr(arg) if ctx.mode == 1 else None, ptr(de), S, k, D, ctx.mode, stream())
        return de, None


# ---- scorers ----------------------------------------------------------------------------------------------------------------
class ScoreDotFn(Function):
    """einsum('be,bce->bc') — sgd_alg.py:2114."""

    @staticmethod
    def forward(ctx, u, i):
        _need_cuda(u, i)
        u, i = _f32c(u), _f32c(i)
        B, N, D = i.shape
        out = torch.empty(B, N, device=u.device, dtype=torch.float32)
        call('sbr_score_dot_fwd', ptr(u), ptr(i), ptr(out), B, N, D, stream())
        ctx.save_for_backward(u, i)
        return out

    @staticmethod
    def backward(ctx, g):
        u, i = ctx.saved_tensors
        g = _f32c(g)
        B, N, D = i.shape
        du = torch.empty_like(u) if ctx.needs_input_grad[0] else None
        di = torch.empty_like(i) if ctx.needs_input_grad[1] else None
        call('sbr_score_dot_bwd', ptr(g), ptr(u), ptr(i), ptr(du), ptr(di), B, N, D, stream())
        return du, di


class LookupFn(Function):
    """nn.Embedding (dense gradient) — sgd_alg.py:144-145, 159-167: out[..., :] = W[idx[...], :]; dW is the dense scatter-add."""

    @staticmethod
    def forward(ctx, W, idx):
        _need_cuda(W, idx)
        if W.stride(-1) != 1:
            W = W.contiguous()
        rows = idx.reshape(-1).to(torch.int32).contiguous()
        n, D = rows.numel(), W.shape[1]
        out = torch.empty(n, D, device=W.device, dtype=torch.float32)
        call('sbr_gather_rows', ptr(W), W.stride(0), ptr(rows), ptr(out), D, None, n, D, stream())
        ctx.save_for_backward(rows)
        ctx.w_shape = W.shape
        return out.view(*idx.shape, D)

    @staticmethod
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        D = ctx.w_shape[1]
        g = _f32c(g).reshape(-1, D)
        dW = torch.zeros(ctx.w_shape, device=g.device, dtype=torch.float32)
        call('sbr_scatter_add_rows', ptr(g), D, None, ptr(rows), ptr(dW), D, rows.numel(), D, stream())
        return dW, None


class BiasScoreFn(Function):
    """out[b, n] = base[b, n] + user_bias[u[b]] + item_bias[i[b, n]] + global_bias (sgd_alg.py:186-194, 110-119); every term
    optional (None). Bias tables are 1-D float views of the [n, 1] embedding weights. u None: row b; i None: column n."""

    @staticmethod
    def forward(ctx, base, ub, ib, gb, u, i, B, N):
        dev = next(t for t in (base, ub, ib, gb) if t is not None).device
        base_c = _f32c(base) if base is not None else None
        u_c = u.long().contiguous() if u is not None else None
        i_c = i.long().contiguous() if i is not None else None
        out = torch.empty(B, N, device=dev, dtype=torch.float32)
        call('sbr_bias_score_add_fwd', ptr(ub), ptr(ib), ptr(gb), ptr(u_c), ptr(i_c), ptr(base_c), ptr(out), B, N, stream())
        ctx.save_for_backward(u_c, i_c)
        ctx.shapes = (None if ub is None else ub.shape, None if ib is None else ib.shape, gb is not None, base is not None, B, N)
        return out

    @staticmethod
    def backward(ctx, g):
        u_c, i_c = ctx.saved_tensors
        ub_s, ib_s, has_gb, has_base, B, N = ctx.shapes
        g = _f32c(g)
        need = ctx.needs_input_grad
        d_ub = torch.zeros(ub_s, device=g.device, dtype=torch.float32) if ub_s is not None and need[1] else None
        d_ib = torch.zeros(ib_s, device=g.device, dtype=torch.float32) if ib_s is not None and need[2] else None
        d_gb = torch.zeros(1, device=g.device, dtype=torch.float32) if has_gb and need[3] else None
        if d_ub is not None or d_ib is not None or d_gb is not None:
            call('sbr_bias_score_bwd', ptr(g), ptr(u_c), ptr(i_c), ptr(d_ub), ptr(d_ib), ptr(d_gb), B, N, stream())
        return (g if has_base and need[0] else None), d_ub, d_ib, d_gb, None, None, None, None


class ScoreAllFn(Function):
    """einsum('be,ce->bc') — sgd_alg.py:2109: all users of the batch against all item representations."""

    @staticmethod
    def forward(ctx, u, i):
        _need_cuda(u, i)
        u, i = _f32c(u), _f32c(i)
        ctx.save_for_backward(u, i)
        return linear_nt(u, i)

    @staticmethod
    def backward(ctx, g):
        u, i = ctx.saved_tensors
        g = _f32c(g)
        du = matmul_nn(g, i) if ctx.needs_input_grad[0] else None
        di = matmul_tn(g, u) if ctx.needs_input_grad[1] else None
        return du, di


# ---- losses -------------------------------------------------------------------------------------------------------------------
LOSS_CODES = {'bce': 0, 'bpr': 1, 'sampled_softmax': 2}


class RecLossFn(Function):
    """train/rec_losses.py:43-113. Returns a float64 scalar for bce / bpr (float64 labels promote the computation in the
    reference) and a float32 scalar for sampled softmax."""

    @staticmethod
    def forward(ctx, logits, labels, kind: int, scale: float, shift: float):
        _need_cuda(logits)
        logits = _f32c(logits)
        B, N = logits.shape
        lab = None
        if kind != 2:
            lab = labels.to(device=logits.device, dtype=torch.float64).contiguous()
        out = torch.empty((), device=logits.device, dtype=torch.float64)
        call('sbr_rec_loss_fwd', kind, ptr(logits), ptr(lab), B, N, scale, shift, ptr(out), stream())
        ctx.args = (kind, scale, shift)
        ctx.save_for_backward(logits, lab if lab is not None else torch.empty(0, device=logits.device))
        return out if kind != 2 else out.float()

    @staticmethod
    def backward(ctx, g):
        logits, lab = ctx.saved_tensors
        kind, scale, shift = ctx.args
        B, N = logits.shape
        g = g.contiguous()
        d = torch.empty_like(logits)
        call('sbr_rec_loss_bwd', kind, ptr(logits), ptr(lab) if kind != 2 else None, B, N, scale, shift, ptr(g),
             1 if g.dtype == torch.float64 else 0, ptr(d), stream())
        return d, None, None, None, None


_INFONCE_WS = {}


def infonce_uses_gemm(G: int, N: int) -> bool:
    """Large groups (in-batch contrast) go through the MFMA GEMMs; many small groups stay in the one-workgroup-per-group
    kernel whose logits live in LDS."""
    return N > infonce_max_n() or (N > 32 and G < 32)


def _infonce_ws(device, N, D):
    from ._lib import lib
    need = lib().sbr_infonce_gemm_workspace(N, D)
    ws = _INFONCE_WS.get(device)
    if ws is None or ws.numel() < need:
        ws = _grow(_INFONCE_WS, device, need, lambda n: torch.empty(n, device=device, dtype=torch.uint8))
    return ws


def infonce_fwd(a_ptr, b_ptr, ld, G, N, D, tau, scale, loss_out, device):
    if infonce_uses_gemm(G, N):
        ws = _infonce_ws(device, N, D)
        call('sbr_infonce_gemm_fwd', a_ptr, b_ptr, ld, G, N, D, tau, scale, ptr(loss_out), ptr(ws), ws.numel(), stream())
    else:
        call('sbr_infonce_fwd', a_ptr, b_ptr, ld, G, N, D, tau, scale, ptr(loss_out), stream())


def infonce_bwd(a_ptr, b_ptr, ld, G, N, D, tau, scale, gout, da_ptr, db_ptr, ldg, device):
    if infonce_uses_gemm(G, N):
        ws = _infonce_ws(device, N, D)
        call('sbr_infonce_gemm_bwd', a_ptr, b_ptr, ld, G, N, D, tau, scale, ptr(gout), da_ptr, db_ptr, ldg, ptr(ws), ws.numel(),
             stream())
    else:
        call('sbr_infonce_bwd', a_ptr, b_ptr, ld, G, N, D, tau, scale, ptr(gout), da_ptr, db_ptr, ldg, stream())


class InfoNCEFn(Function):
    """train/regularization_losses.py:14-43 on two [G, N, D] views that may be strided slices of one [G*N, 2, D] tensor."""

    @staticmethod
    def forward(ctx, e, tau: float, mean: bool, G: int, N: int):
        # e: [G*N, 2, D] contiguous; a = e[:, 0], b = e[:, 1]
        _need_cuda(e)
        e = _f32c(e)
        D = e.shape[-1]
        out = torch.empty((), device=e.device, dtype=torch.float64)
        scale = 1.0 / (G * N) if mean else 1.0
        infonce_fwd(e[:, 0].data_ptr(), e[:, 1].data_ptr(), 2 * D, G, N, D, tau, scale, out, e.device)
        ctx.args = (tau, scale, G, N, D)
        ctx.save_for_backward(e)
        return out.float()

    @staticmethod
    def backward(ctx, g):
        (e,) = ctx.saved_tensors
        tau, scale, G, N, D = ctx.args
        g = g.float().contiguous()
        de = torch.empty_like(e)
        infonce_bwd(e[:, 0].data_ptr(), e[:, 1].data_ptr(), 2 * D, G, N, D, tau, scale, g, de[:, 0].data_ptr(),
                    de[:, 1].data_ptr(), 2 * D, e.device)
        return de, None, None, None, None


def infonce_max_n() -> int:
    from ._lib import lib
    return lib().sbr_infonce_max_n()


# ---- evaluation helpers -----------------------------------------------------------------------------------------------------
def mask_scores_(scores: torch.Tensor, u_idx: torch.Tensor, excl_indptr: torch.Tensor, excl_indices: torch.Tensor, item_offset: int = None):
    """``item_offset`` (item-sharded evaluation): ``scores`` holds the item columns [item_offset, item_offset + scores.shape[1])."""
    _need_cuda(scores)
    if item_offset is None:
        call('sbr_mask_scores', ptr(scores), scores.stride(0), ptr(u_idx), ptr(excl_indptr), ptr(excl_indices), scores.shape[0], stream())
    else:
        call('sbr_mask_scores_shard', ptr(scores), scores.stride(0), ptr(u_idx), ptr(excl_indptr), ptr(excl_indices), scores.shape[0],
             int(item_offset), int(scores.shape[1]), stream())
    return scores


def topk_rows(scores: torch.Tensor, k: int):
    _need_cuda(scores)
    Bu, I = scores.shape
    val = torch.empty(Bu, k, device=scores.device, dtype=torch.float32)
    idx = torch.empty(Bu, k, device=scores.device, dtype=torch.int32)
    call('sbr_topk_rows', ptr(scores), scores.stride(0), Bu, I, k, ptr(val), ptr(idx), stream())
    return val, idx


def rank_metrics(topk_idx: torch.Tensor, u_idx, label_indptr, label_indices, ks: Sequence[int]):
    _need_cuda(topk_idx)
    Bu, kmax = topk_idx.shape
    import ctypes
    ks_arr = (ctypes.c_int * len(ks))(*ks)
    out = torch.empty(3, len(ks), Bu, device=topk_idx.device, dtype=torch.float32)
    call('sbr_rank_metrics', ptr(topk_idx), kmax, ptr(u_idx), ptr(label_indptr), ptr(label_indices), Bu,
         ctypes.cast(ks_arr, ctypes.c_void_p), len(ks), ptr(out), stream())
    return out


def cast_f16(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    x = _f32c(x)
    y = torch.empty(x.shape, device=x.device, dtype=torch.float16)
    call('sbr_cast_f32_to_f16', ptr(x), ptr(y), x.numel(), stream())
    return y


class ScorerExclusions:
    """The exclusion mask of one (user list, exclusion CSR, item range) combination in the layout the fused scorer reads (a
    wave-uniform event stream, csrc/score_topk_f16_n.hip). The mask of an evaluation split never changes (eval/eval.py:219:
    ``dataset.exclude_data``), so ``evaluation.evaluate_recommender_algorithm`` keeps one of these per (split, user chunk, item
    shard) next to the resident CSR and every evaluation after the first skips the three builder launches. The first
    ``score_topk_f16`` call that receives the object builds the stream; later calls must pass the same u_idx / CSR / item range / D."""

    def __init__(self):
        self.buf, self.key = None, None


def score_topk_route(route: int = -1) -> int:
    """Which fused scorer ``score_topk_f16`` runs (process-wide; returns the previous setting): 0 automatic — the two-pass scorer
    (csrc/score_topk_f16_2p.hip) for catalogues of >= 8,192 items, the one-pass kernel below —, 1 always one-pass, 2 two-pass or an
    error; -1 only queries. Both return the same lists bit for bit; the switch exists for tests and A/B timing."""
    return int(lib().sbr_score_topk_f16_route(int(route)))


def score_topk_f16(u16: torch.Tensor, i16: torch.Tensor, k: int, u_idx=None, excl_indptr=None, excl_indices=None,
                   item_offset: int = 0, exclusions: 'ScorerExclusions' = None):
    _need_cuda(u16, i16)
    Bu, D = u16.shape
    I = i16.shape[0]
    val = torch.empty(Bu, k, device=u16.device, dtype=torch.float32)
    idx = torch.empty(Bu, k, device=u16.device, dtype=torch.int32)
    nnz = 0 if excl_indices is None else int(excl_indices.numel())
    ws = torch.empty(max(int(lib().sbr_score_topk_f16_workspace(Bu, I, k)), 8), device=u16.device, dtype=torch.uint8)
    ev, build = None, 1
    if nnz > 0:
        key = (Bu, I, D, int(item_offset), nnz)
        holder = exclusions if exclusions is not None else ScorerExclusions()
        if holder.buf is not None and holder.key == key:
            build = 0                          # built by an earlier call for the same users / mask / item range (the caller's promise)
        else:
            holder.buf = torch.empty(int(lib().sbr_score_topk_f16_events_bytes(Bu, nnz)) + 16, device=u16.device, dtype=torch.uint8)
            holder.key = key
        ev = holder.buf
    _timed(('score_topk_f16', Bu, I, D, k),
           lambda: call('sbr_score_topk_f16', ptr(u16), ptr(i16), D, Bu, I, ptr(u_idx), ptr(excl_indptr), ptr(excl_indices), nnz,
                        item_offset, k, ptr(val), ptr(idx), ptr(ws), ws.numel(), ptr(ev), 0 if ev is None else ev.numel(), build, stream()))
    return val, idx


# ---- optimizer steps ----------------------------------------------------------------------------------------------------------
def adam_step(kind: int, p, g, m, v, lr, b1, b2, eps, wd, step: int, zero_grad: bool = False, copy=None):
    """One dense Adam / AdamW step; ``zero_grad``: the gradient is reset by the same launch (step() + zero_grad());
    ``copy`` = (src, dst) float64 tensors of <= 256 elements (needs ``zero_grad``): copied by the same launch."""
    if zero_grad:
        src, dst = copy if copy is not None else (None, None)
        call('sbr_adam_step_zero_grad', kind, ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, b1, b2, eps, wd, step, ptr(src), ptr(dst),
             0 if src is None else src.numel(), stream())
    else:
        assert copy is None
        call('sbr_adam_step', kind, ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, b1, b2, eps, wd, step, stream())


def adagrad_step(p, g, s, lr, eps, wd):
    call('sbr_adagrad_step', ptr(p), ptr(g), ptr(s), p.numel(), lr, eps, wd, stream())
