"""Thin torch-tensor -> C-ABI adapters (plumbing only: device pointers, strides, the current HIP stream).

Every function here enqueues hand-written gfx950 kernels from libttsk_hip.so on torch's current stream and
returns immediately; nothing is computed by PyTorch.  Tensors must live on a HIP device ("cuda:N" in
PyTorch-ROCm naming) — CPU tensors are rejected, there is no fallback.
"""
import ctypes as C

import threading

import torch

from . import lib as L
from .lib import (A_TR, B_TR, C_F32, RELU, ADD_R, R_F32, MASK_G, LRELU_IN, TANH, ACCUM_C, LRELU_OUT, F16, C2_LRELU, DEFER_REDUCE, RAW_SLABS,  # noqa: F401
                  GemmDesc, check)

bf16 = torch.bfloat16
f16 = torch.float16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.TtskError("tts_king_amd kernels need HIP device tensors (got %s); there is no CPU path" % t.device)


def _ptr(t):
    return None if t is None else t.data_ptr()


import os as _os
# K steps one workgroup of a grouped weight-gradient launch walks (128x128 tile, 256x128 tile): more steps = fewer split-K slabs
DW_STEPS_PER_WG = (28, 72)
LAUNCH_COUNTS = None   # bench.py sets this to a dict: grouped-GEMM / batched-reducer / column-sum launches issued by the flushes
STAMPS = None       # tools/debug/step_stamps.py (diagnostic library only): (uint64 device buffer, [names]) — `stamp(name)` then launches a
                    # one-thread kernel that writes the device's 100 MHz clock into the next slot, on the current stream
GEMM_TRACE = None   # bench.py sets this to a list: every ttsk_gemm launch is then bracketed by HIP events on its stream
                    # (and, through `_family`, every launch of the other MFMA kernel families and of the step's HBM-bound passes)


def _family(kind, flops):
    """Decorator: when GEMM_TRACE is set, bracket the wrapped launch with HIP events on its stream and record (events, algorithmic
    FLOPs = flops(*args, **kwargs), family name) — bench.py's per-family roofline (`roofline.families`).  No cost otherwise."""
    def deco(fn):
        import functools

        @functools.wraps(fn)
        def wrapped(*a, **kw):
            if GEMM_TRACE is None:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            GEMM_TRACE.append((e0, e1, float(flops(*a, **kw)), kind, (0, 0, 0, 1, 1, 0, fn.__name__)))      # (last: the wrapper, for bench.py's per-instance rows)
            return out
        return wrapped
    return deco


def _conv_flops(x, packed, Cout, k, *a, **kw):
    return 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * Cout * k


def stamp(name):
    """Diagnostic (a no-op unless ops.STAMPS is set, which needs libttsk_hip_stamps.so): mark this point of the current stream."""
    if STAMPS is None:
        return
    buf, names = STAMPS
    i = len(names)
    if i >= buf.numel():
        return
    names.append(name)
    fn = L.load().ttsk_debug_stamp
    fn.argtypes = [C.c_void_p, C.c_void_p]
    fn(C.c_void_p(buf.data_ptr() + 8 * i), C.c_void_p(_stream()))


class Slabs:
    """Raw split-K partial tiles of a GEMM (gemm(raw=True)): fp32 [splits][M*N] in `ws`."""

    def __init__(self, ws, splits, stride):
        self.ws, self.splits, self.stride = ws, splits, stride


class DeferQueue(list):
    """What `gemm(defer=...)` collects during a backward pass: split-K reduce items (the list itself) and, in `.group`, whole
    weight-gradient GEMMs that run as grouped launches at `flush_deferred` (nothing but the optimiser reads their results)."""

    def __init__(self, group_gemms=True):
        super().__init__()
        self.group = [] if group_gemms else None
        self.dwconv = []        # weight gradients of convs with taps: (dy, x, dst, lens, accumulate) for dwconv_batch (csrc/dwconv.hip)
        self.dwgemm = []        # ... of Linear / k = 1 / PostNet layers with 256-multiple channels: (dy, x, dst, lens, accumulate, splits)


class GemmGroup:
    """Independent GEMMs (same operand layout per launch) that run as one grouped launch: `gemm(..., group=g)` ... `g.flush()`.
    Problems are launched per (operand layout, tile configuration): at most a handful of grids."""

    def __init__(self):
        self.descs, self.keep = [], []

    def flush(self):
        flush_group(self.descs, self.keep)
        self.keep = []


def flush_group(descs, keep, max_wgs=0, upload_only=False):
    """Grouped launches (ttsk_gemm_group_*) of the queued descriptors, one per operand layout; `keep` holds their tensors.
    max_wgs > 0 caps the grid of the 256x128 configuration (ttsk_gemm_group_launch_capped).
    upload_only: only the tables go to the device (current stream); returns a function that issues the launches (on the stream
    current when it is called, which the caller has ordered behind this one)."""
    if not descs:
        return (lambda streams=None: None) if upload_only else None
    pending = []
    lib = L.load()
    by_layout = {}
    for d in descs:
        by_layout.setdefault((d.flags & (A_TR | B_TR | F16), d.kernel), []).append(d)
    dev = keep[0].device
    for ds in by_layout.values():
        # longest workgroups first: a grouped grid is dispatched in workgroup order and its problems differ several-fold in K steps
        # per workgroup (HiFi-GAN's k = 11 / 7 / 3 convs, a decoder vs an encoder weight gradient), so the short ones should fill the
        # tail of the launch instead of opening it.  The problems are independent: their order changes no result.
        ds.sort(key=lambda d: -(d.K * max(d.taps, 1)) // max(d.splits, 1))
        n = len(ds)
        nbytes = int(lib.ttsk_gemm_group_table_bytes(n))
        host = (C.c_ubyte * nbytes)()
        arr = (GemmDesc * n)(*ds)
        total = C.c_int32(0)
        check(lib.ttsk_gemm_group_build(arr, n, host, C.byref(total)), "ttsk_gemm_group_build")
        table = torch.empty(nbytes, dtype=torch.uint8, device=dev)      # filled by group_launch through kernel arguments
        if upload_only:
            check(lib.ttsk_gemm_group_upload(host, C.c_void_p(table.data_ptr()), _stream()), "ttsk_gemm_group_upload")
            fl = sum(2.0 * d.M * d.N * d.K * max(d.taps, 1) * d.nz1 * d.nz2 for d in ds)
            kind = ("TT" if ds[0].flags & A_TR else ("NT_btr" if ds[0].flags & B_TR else "NT")) + "%dg" % ds[0].kernel
            pending.append((host, table, fl, kind, (n, 0, 0, 1, 1, int(total.value))))
            keep.append(table)
            if LAUNCH_COUNTS is not None:
                LAUNCH_COUNTS["grouped_gemm"] = LAUNCH_COUNTS.get("grouped_gemm", 0) + 1
            continue
        if GEMM_TRACE is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(lib.ttsk_gemm_group_launch_capped(host, C.c_void_p(table.data_ptr()), int(max_wgs), _stream()), "ttsk_gemm_group_launch")
        if LAUNCH_COUNTS is not None:
            LAUNCH_COUNTS["grouped_gemm"] = LAUNCH_COUNTS.get("grouped_gemm", 0) + 1
        if GEMM_TRACE is not None:
            e1.record()
            fl = sum(2.0 * d.M * d.N * d.K * max(d.taps, 1) * d.nz1 * d.nz2 for d in ds)
            kind = "TT" if ds[0].flags & A_TR else ("NT_btr" if ds[0].flags & B_TR else "NT")
            GEMM_TRACE.append((e0, e1, fl, kind + "%dg" % ds[0].kernel, (n, 0, 0, 1, 1, int(total.value))))
        keep.append(table)
    descs.clear()
    if upload_only:
        def launch(streams=None):
            """Issue the grouped launches whose tables are uploaded.  `streams`: torch streams (each already ordered behind the
            uploads by the caller) to spread the launches over, one per launch in turn — they are independent problems."""
            for i, (host, table, fl, kind, shape) in enumerate(pending):
                ctx = torch.cuda.stream(streams[i % len(streams)]) if streams else None
                if ctx is not None:
                    ctx.__enter__()
                try:
                    _launch_uploaded(host, table, fl, kind, shape)
                finally:
                    if ctx is not None:
                        ctx.__exit__(None, None, None)

        def _launch_uploaded(host, table, fl, kind, shape):
            if GEMM_TRACE is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(lib.ttsk_gemm_group_launch_uploaded(host, C.c_void_p(table.data_ptr()), int(max_wgs), _stream()), "ttsk_gemm_group_launch_uploaded")
            if GEMM_TRACE is not None:
                e1.record()
                GEMM_TRACE.append((e0, e1, fl, kind, shape))
        return launch


def upload_deferred_gemms(items, max_wgs=0, with_dwconv=True, with_dwgemm=True):
    """flush_deferred_gemms in two parts: the tables now (current stream), the launches when the returned function is called (the
    queued dwconv problems carry their table in the launch's arguments: with_dwconv, they go with the launches)."""
    group = getattr(items, "group", None)
    launch = flush_group(group, getattr(items, "_keep"), max_wgs, upload_only=True) if group else (lambda streams=None: None)

    def both(streams=None):
        if with_dwconv:
            flush_dwconv(items)
        if with_dwgemm:
            flush_dwgemm(items)
        launch(streams)
    return both


def flush_dwconv(items, n=None):
    """The (first n) queued conv weight gradients as ttsk_dwconv_batch launches (12 problems each, one kernel size each) on the
    current stream.  Their operands stay alive in the queue's keep list until its final flush."""
    q = getattr(items, "dwconv", None)
    if not q:
        return
    now = q if n is None else q[:n]
    rest = [] if n is None else q[n:]
    if not hasattr(items, "_keep"):
        items._keep = []
    by_k = {}
    for it in now:
        by_k.setdefault(it[2].shape[1], []).append(it)
        items._keep.extend(t for t in it[:4] if t is not None)
    for lst in by_k.values():
        for i in range(0, len(lst), 12):
            dwconv_batch(lst[i:i + 12])
    q[:] = rest


DWG_TARGET_STEPS = (72, 72)   # K steps per workgroup: k = 1, taps (3 ranges at B = 16, T = 423: the measured optimum of 2 / 3 / 4)


DWG_SHORT_STEPS = 16


def dwgemm_splits(Bsz, S, k=1):
    """Utterance ranges per problem for dwgemm_batch: workgroups of about DWG_TARGET_STEPS 32-row K steps — at the 16 x 423-row step 4
    ranges for every problem (432 workgroups; 3 ranges for the PostNet's k = 5 — 372 workgroups, two even rounds of the capped 192 —
    measured the same: 2.816 vs 2.804 ms); short problems (the phoneme side: 2 steps per utterance) of about 16 steps, so that their few
    tiles still spread over the chip."""
    steps = Bsz * ((S + 31) // 32)
    target_steps = DWG_TARGET_STEPS[0] if k == 1 else DWG_TARGET_STEPS[1]
    if steps <= 64:
        target_steps = DWG_SHORT_STEPS
    return max(1, min(Bsz, (steps + target_steps // 2) // target_steps))


def flush_dwgemm(items, reduce_now=False, max_wgs=0):
    """The queued dwgemm problems as ttsk_dwgemm_batch launches on the current stream; the reducer items of the split ones join the
    queue's split-K items (summed by flush_deferred's batched reducer), or are summed right here (reduce_now)."""
    q = getattr(items, "dwgemm", None)
    if not q:
        return
    if not hasattr(items, "_keep"):
        items._keep = []
    for it in q:
        items._keep.extend(t for t in it[:4] if t is not None)
    red = dwgemm_batch(q, max_wgs)
    q[:] = []
    if reduce_now and red:
        arr = (L.ReduceItem * len(red))(*[r for r, _ in red])
        check(L.load().ttsk_gemm_reduce_batch(arr, len(red), _stream()), "ttsk_gemm_reduce_batch")
        items._keep.extend(ws for _, ws in red)
        if LAUNCH_COUNTS is not None:
            LAUNCH_COUNTS["reduce_batch"] = LAUNCH_COUNTS.get("reduce_batch", 0) + 1
    else:
        items.extend(red)


def queue_dw(defer, dy, x, dst, lens, accumulate, k=1, use_dwgemm=True):
    """Queue the weight gradient dst (Cout, k, Cin) (+)= dy (B,S,Cout)^T x (B,S,Cin) [taps: "same" padding]: on the 256x256-tile kernel
    (dwgemm.hip) when the shape allows, else as a grouped GEMM.  `lens`: rows of each utterance that carry a gradient (or None)."""
    Bsz, S, Cout = dy.shape
    Cin = x.shape[2]
    if (use_dwgemm and getattr(defer, "group", None) is not None and dy.dtype == bf16 and x.dtype == bf16 and dwgemm_supported(Cout, Cin, k)
            and dst.is_contiguous()):
        sp = dwgemm_splits(Bsz, S, k)
        if (Bsz + sp - 1) // sp <= 64:
            defer.dwgemm.append((dy, x, dst.view(Cout, k, Cin), lens, accumulate, sp))
            return
    if k == 1:
        linear_dw(dy.reshape(Bsz * S, Cout), x.reshape(Bsz * S, Cin), dst.view(Cout, Cin), defer=defer, accumulate=accumulate)
    else:
        conv1d_dw(dy, x, dst, k=k, defer=defer, accumulate=accumulate)


def flush_deferred_gemms(items, max_wgs=0, frac=1.0, small_too=False):
    """Only the grouped weight-gradient GEMMs queued in `items` so far, as grouped launches on the current stream (grid capped at
    max_wgs workgroups when > 0); their split-K slabs stay queued for `flush_deferred`'s reducer launch.  frac < 1: only about
    that fraction of the queued FLOPs (the problems queued first); the rest stays queued for the next flush.  The queued dwconv
    and dwgemm problems go first, all of them (capped likewise; with max_wgs their slabs are summed right behind them: this is the
    side stream's launch, which has the time)."""
    flush_dwconv(items)
    stamp("dw.dwconv")
    flush_dwgemm(items, reduce_now=max_wgs > 0, max_wgs=max_wgs)
    stamp("dw.dwgemm")
    group = getattr(items, "group", None)
    if not group:
        return
    if max_wgs > 0:
        # a capped launch exists for the 256x128 configuration only; the few 128x128 problems (80-row / 80-column outputs: mel_linear, the
        # PostNet's first and last conv) go behind it uncapped (small_too: 168 workgroups, two per CU) or stay queued for the final flush
        small = [d for d in group if d.kernel != 2]
        if small:
            group[:] = [d for d in group if d.kernel == 2]
            try:
                if group:
                    flush_deferred_gemms(items, max_wgs, frac)
            finally:
                group.extend(small)
            if small_too:
                now = [d for d in group if d.kernel != 2]
                group[:] = [d for d in group if d.kernel == 2]
                flush_group(now, getattr(items, "_keep"), 0)
            return
    if frac < 1.0:
        fl = [2.0 * d.M * d.N * d.K * max(d.taps, 1) * d.nz1 * d.nz2 for d in group]
        want, acc, n = frac * sum(fl), 0.0, 0
        while n < len(group) and acc < want:
            acc += fl[n]
            n += 1
        now, later = group[:n], group[n:]
        flush_group(now, getattr(items, "_keep"), max_wgs)
        group[:] = later
        return
    flush_group(group, getattr(items, "_keep"), max_wgs)


def flush_deferred_prefix(items, n_group, n_reduce, max_wgs=0, n_dwconv=0):
    """The first `n_dwconv` queued dwconv problems, the first `n_group` queued weight-gradient GEMMs as grouped launches (grid capped at
    max_wgs when > 0) and then the batched reducer for the first `n_reduce` queued split-K items, on the current stream; all are
    removed from the queue.  The data-parallel "side" schedule flushes the queue bucket by bucket this way
    (FastSpeech2._launch_dw_side_buckets)."""
    if n_dwconv > 0:
        flush_dwconv(items, n_dwconv)
    group = getattr(items, "group", None)
    if group and n_group > 0:
        now = group[:n_group]
        del group[:n_group]
        flush_group(now, getattr(items, "_keep"), max_wgs)
    if n_reduce > 0:
        head = items[:n_reduce]
        del items[:n_reduce]
        arr = (L.ReduceItem * len(head))(*[it for it, _ in head])
        check(L.load().ttsk_gemm_reduce_batch(arr, len(head), _stream()), "ttsk_gemm_reduce_batch")
        if LAUNCH_COUNTS is not None:
            LAUNCH_COUNTS["reduce_batch"] = LAUNCH_COUNTS.get("reduce_batch", 0) + 1
        if hasattr(items, "_keep"):
            items._keep.extend(ws for _, ws in head)       # the slabs stay alive until the queue's final flush


def flush_deferred(items):
    """The queued weight-gradient GEMMs as grouped launches, then one ttsk_gemm_reduce_batch launch (per 64 items) for the
    split-K slabs collected in `items` (see gemm(defer=...))."""
    flush_dwconv(items)
    flush_dwgemm(items)
    group = getattr(items, "group", None)
    if group:
        keep = getattr(items, "_keep")
        flush_group(group, keep)
    if not items:
        if group is not None:
            items._keep = []
        return
    arr = (L.ReduceItem * len(items))(*[it for it, _ in items])
    check(L.load().ttsk_gemm_reduce_batch(arr, len(items), _stream()), "ttsk_gemm_reduce_batch")
    if LAUNCH_COUNTS is not None:
        LAUNCH_COUNTS["reduce_batch"] = LAUNCH_COUNTS.get("reduce_batch", 0) + 1
    items.clear()
    if group is not None:
        items._keep = []


def plan(d):
    """(kernel, splits, workspace_bytes) the library will use for descriptor `d` (ttsk_gemm_plan)."""
    k, sp, ws = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    check(L.load().ttsk_gemm_plan(C.byref(d), C.byref(k), C.byref(sp), C.byref(ws)), "ttsk_gemm_plan")
    return k.value, sp.value, ws.value


def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, flags=0, alpha=1.0, bias=None, R=None, ldr=0, G=None, ldg=0, C2=None,
         nz1=1, nz2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), taps=0, seg_len=0, tap_shift0=0, tap_dshift=0,
         b_tap_stride=0, bseg_len=0, bshift0=0, bdshift=0, out_seg=0, out_mul=0, out_add=0, out_add_dz=0, splits=0, kernel=0,
         in_slope=0.0, out_slope=0.0, defer=None, group=None, s_bias1=0, raw=False):
    """Raw descriptor-level call of ttsk_gemm (see include/ttsk.h).  A/B/Cout may be views: the data pointer of
    the view is the operand origin.  splits / kernel = 0 let the library plan (tile configuration, split-K factor);
    the split-K workspace is allocated here (the C library never allocates).  `defer`: a list — a split-K weight-
    gradient GEMM then leaves its slabs un-reduced and appends a reduce item to it (see flush_deferred).  `group`: a
    GemmGroup — independent problems collected there run as ONE grouped launch at `group.flush()`.  `raw`: leave the fp32
    partial tiles [splits][M][N] un-reduced and without epilogue (TTSK_GEMM_RAW_SLABS); returns Slabs(ws, splits, M*N) for
    layernorm_bwd(slabs=...) — `Cout` may be None."""
    _dev(A, B, Cout, bias, R, G, C2)
    if raw:
        flags |= RAW_SLABS | C_F32
        if Cout is None:
            Cout = A                  # not written: any valid device pointer
    d = GemmDesc()
    d.A, d.B, d.C, d.C2 = _ptr(A), _ptr(B), _ptr(Cout), _ptr(C2)
    d.bias, d.R, d.G = _ptr(bias), _ptr(R), _ptr(G)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldr, d.ldg = lda, ldb, ldc, ldr, ldg
    if A.dtype == f16:
        flags |= F16
    if Cout.dtype == torch.float32 and not raw:
        flags |= C_F32
    if R is not None:
        flags |= ADD_R | (R_F32 if R.dtype == torch.float32 else 0)
    if G is not None:
        flags |= MASK_G
    d.flags, d.alpha, d.in_slope, d.out_slope = flags, alpha, in_slope, out_slope
    d.nz1, d.nz2 = nz1, nz2
    d.sA1, d.sA2 = sA
    d.sB1, d.sB2 = sB
    d.sC1, d.sC2 = sC
    d.sR1, d.sR2 = sR
    d.taps, d.seg_len, d.tap_shift0, d.tap_dshift, d.b_tap_stride = taps, seg_len, tap_shift0, tap_dshift, b_tap_stride
    d.bseg_len, d.bshift0, d.bdshift = bseg_len, bshift0, bdshift
    d.out_seg, d.out_mul, d.out_add, d.out_add_dz = out_seg, out_mul, out_add, out_add_dz
    d.splits, d.kernel, d.s_bias1 = splits, kernel, s_bias1
    user_splits, user_kernel = splits, kernel
    kernel, splits, ws_bytes = plan(d)
    if kernel == 3 and group is not None:
        # grouped launches exist for the 128x128 and 256x128 tiles only: plan again with the 128x128 tile
        d.kernel, d.splits = 1, user_splits
        kernel, splits, ws_bytes = plan(d)
    grouped_dw = (defer is not None and getattr(defer, "group", None) is not None and nz1 == 1 and (flags & C_F32)
                  and not (flags & ~(A_TR | B_TR | C_F32 | ACCUM_C)) and bias is None and A.dtype != f16)
    if grouped_dw and user_splits == 0 and user_kernel == 0:
        # The planner splits K until ONE problem fills the chip (a 256x256 weight gradient over 6768 rows: 36 ways, 36 fp32
        # slabs to write and to reduce).  In a grouped launch the other problems fill it: ~28 (128x128 tiles) / ~36 (256x128) K steps per workgroup keep the
        # grid balanced with 4x-9x fewer slabs (750 MB -> ~50 MB of split-K traffic per train step).
        # Tile configuration: the 256x128 LDS-DMA kernel unless the output has too few rows to fill its tile.
        steps = (K + 63) // 64 * max(taps, 1)
        d.kernel = 2 if (M >= 192 and N >= 128) else 1        # (an 80-column output on the 256x128 tile would waste 3/8 of it)
        per1, per2 = DW_STEPS_PER_WG
        d.splits = max(1, (steps + per1 // 2) // per1) if d.kernel == 1 else max(1, (steps + per2 // 2) // per2)
        kernel, splits, ws_bytes = plan(d)
    d.splits, d.kernel = splits, kernel
    ws = None
    if ws_bytes > 0:
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=A.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws_bytes
        if defer is not None and nz1 == 1 and (flags & C_F32) and not (flags & ~(A_TR | B_TR | C_F32 | ACCUM_C)) and bias is None:
            d.flags = flags | DEFER_REDUCE
            it = L.ReduceItem()
            it.ws, it.C, it.M, it.N, it.ldc, it.nz, it.splits = ws.data_ptr(), Cout.data_ptr(), M, N, ldc, nz2, splits
            it.accumulate, it.sC2, it.alpha = int(bool(flags & ACCUM_C)), sC[1], alpha
            defer.append((it, ws))
    if raw:
        check(L.load().ttsk_gemm(C.byref(d), _stream()), "ttsk_gemm")
        return Slabs(ws, splits, M * N * nz1 * nz2)
    if group is not None:
        group.descs.append(d)
        group.keep.extend(t for t in (A, B, Cout, ws, bias, R, G, C2) if t is not None)
        return Cout
    if grouped_dw:
        # a weight-gradient GEMM: queued whole, launched with the others at flush_deferred
        defer.group.append(d)
        if not hasattr(defer, "_keep"):
            defer._keep = []
        defer._keep.extend(t for t in (A, B, Cout, ws) if t is not None)
        return Cout
    if GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(L.load().ttsk_gemm(C.byref(d), _stream()), "ttsk_gemm")
        e1.record()
        kind = "TT" if flags & A_TR else ("NT_btr" if flags & B_TR else "NT")
        GEMM_TRACE.append((e0, e1, 2.0 * M * N * K * max(taps, 1) * nz1 * nz2, kind + str(kernel), (M, N, K, max(taps, 1), nz1 * nz2, splits)))
        return Cout
    check(L.load().ttsk_gemm(C.byref(d), _stream()), "ttsk_gemm")
    return Cout


_DUR_DTYPE = {torch.int64: 0, torch.float32: 1, torch.int32: 2}


def length_regulator_fwd(x, dur, T, pe=None, want_idx=True):
    """x (B,L,D) bf16, dur (B,L) int64/fp32/int32 -> out (B,T,D) bf16, idx (B,T) int32, cumsum (B,L) int32,
    mel_len (B,) int64 (uncropped).  reference: fs_two/model/modules.py:225-252."""
    _dev(x, dur, pe)
    B, Lp, D = x.shape
    out = torch.empty(B, T, D, dtype=bf16, device=x.device)
    idx = torch.empty(B, T, dtype=torch.int32, device=x.device) if want_idx else None
    cs = torch.empty(B, Lp, dtype=torch.int32, device=x.device)
    mel_len = torch.empty(B, dtype=torch.int64, device=x.device)
    dur = dur.contiguous()
    check(L.load().ttsk_length_regulator_fwd(_ptr(x.contiguous()), _ptr(dur), _DUR_DTYPE[dur.dtype], _ptr(pe), _ptr(out),
                                             _ptr(idx), _ptr(cs), _ptr(mel_len), B, Lp, T, D, _stream()),
          "ttsk_length_regulator_fwd")
    return out, idx, cs, mel_len


def length_regulator_bwd(dout, cs, L_src):
    _dev(dout, cs)
    B, T, D = dout.shape
    dx = torch.empty(B, L_src, D, dtype=bf16, device=dout.device)
    check(L.load().ttsk_length_regulator_bwd(_ptr(dout.contiguous()), _ptr(cs), _ptr(dx), B, L_src, T, D, _stream()),
          "ttsk_length_regulator_bwd")
    return dx


# ----------------------------------------------------------------------------------------------------------------
# GEMM-shaped ops expressed on the one kernel.  Activations are channels-last bf16 [rows][C], rows = B*T.
# Weights are the bf16 shadows: Linear (out,in); Conv1d (out, k, in)  [the fp32 masters use the same layout].

def linear(x, W, bias=None, out=None, flags=0, out_dtype=None, R=None, G=None, C2=None, alpha=1.0, out_slope=0.0, **kw):
    """y[M,N] = x[M,K] @ W[N,K]^T (+bias, epilogue).  reference: nn.Linear sites SubLayers.py:41-43,62; fastspeech2.py:102."""
    M, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or x.dtype, device=x.device)
    return gemm(x, W, out, M, N, K, x.stride(0), W.stride(0), out.stride(0), flags=flags, alpha=alpha, bias=bias,
                R=R, ldr=0 if R is None else R.stride(0), G=G, ldg=0 if G is None else G.stride(0), C2=C2,
                out_slope=out_slope, **kw)


def linear_dx(dy, W, out=None, R=None, G=None, out_dtype=bf16, **kw):
    """dx[M,K] = dy[M,N] @ W[N,K]   (W read in place through the transposing LDS read)."""
    M, N = dy.shape
    K = W.shape[1]
    if kw.get("raw"):
        return gemm(dy, W, None, M, K, N, dy.stride(0), W.stride(0), K, flags=B_TR, **kw)
    if out is None:
        out = torch.empty(M, K, dtype=out_dtype, device=dy.device)
    return gemm(dy, W, out, M, K, N, dy.stride(0), W.stride(0), out.stride(0), flags=B_TR, R=R,
                ldr=0 if R is None else R.stride(0), G=G, ldg=0 if G is None else G.stride(0), **kw)


def linear_dw(dy, x, dst, accumulate=True, **kw):
    """dst[N,K] (fp32) (+)= dy[M,N]^T @ x[M,K]: contraction over rows (split-K over the rows, deterministic reduce)."""
    M, N = dy.shape
    K = x.shape[1]
    return gemm(dy, x, dst, N, K, M, dy.stride(0), x.stride(0), K, flags=A_TR | B_TR | (ACCUM_C if accumulate else 0), **kw)


def conv1d(x, W, bias, dilation=1, out=None, flags=0, out_dtype=None, R=None, C2=None, in_slope=0.0, out_slope=0.0,
           alpha=1.0, **kw):
    """'same' Conv1d on channels-last activations.  x (B,T,Cin) bf16, W (Cout,k,Cin) bf16 -> (B,T,Cout).
    reference: SubLayers.py:96 (k=9/1), modules.py:337-355 (k=3), Layers.py:59-67 (k=5), hifi/models.py:88-95."""
    Bsz, T, Cin = x.shape
    Cout, k, _ = W.shape
    pad = dilation * (k - 1) // 2
    if out is None:
        out = torch.empty(Bsz, T, Cout, dtype=out_dtype or x.dtype, device=x.device)
    gemm(x, W, out, Bsz * T, Cout, Cin, Cin, k * Cin, Cout, flags=flags, bias=bias, R=R, ldr=Cout, C2=C2, taps=k,
         seg_len=T, tap_shift0=-pad, tap_dshift=dilation, b_tap_stride=Cin, in_slope=in_slope, out_slope=out_slope,
         alpha=alpha, **kw)
    return out


def ffn_conv_supported(Cin, Cout, k):
    return bool(L.load().ttsk_ffn_conv_supported(Cin, Cout, k))


def ffn_pack_weight(W, out=None):
    """Tap-major (Cout,k,256) bf16 -> fragment-major pack for ffn_conv_fwd(packed=...) (ttsk_ffn_pack_weight)."""
    _dev(W, out)
    Cout, k, _ = W.shape
    if out is None:
        out = torch.empty(W.numel(), dtype=bf16, device=W.device)
    check(L.load().ttsk_ffn_pack_weight(_ptr(W), _ptr(out), Cout, k, _stream()), "ttsk_ffn_pack_weight")
    return out


def ffn_pack_weight_batch(Ws, outs):
    """ffn_pack_weight for up to 16 weights of one shape in one launch (ttsk_ffn_pack_weight_batch)."""
    _dev(*Ws, *outs)
    n = len(Ws)
    Cout, k, _ = Ws[0].shape
    src = (C.c_void_p * n)(*[w.data_ptr() for w in Ws])
    dst = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    check(L.load().ttsk_ffn_pack_weight_batch(C.cast(src, C.c_void_p), C.cast(dst, C.c_void_p), n, Cout, k, _stream()), "ttsk_ffn_pack_weight_batch")


def win_conv_supported(Cin, Cout, k):
    return bool(L.load().ttsk_win_conv_supported(Cin, Cout, k))


def win_conv_pack_batch(Ws, outs, transpose=False):
    """Fragment-major packs of up to 16 tap-major weights (Cs, k, Ds) of one shape in one launch (ttsk_win_conv_pack_batch);
    transpose: the pack of the conv's input gradient as a forward conv (taps flipped, channels swapped)."""
    _dev(*Ws, *outs)
    n = len(Ws)
    Cs, k, Ds = Ws[0].shape
    src = (C.c_void_p * n)(*[w.data_ptr() for w in Ws])
    dst = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    check(L.load().ttsk_win_conv_pack_batch(C.cast(src, C.c_void_p), C.cast(dst, C.c_void_p), n, Cs, k, Ds, int(transpose), _stream()),
          "ttsk_win_conv_pack_batch")


def win_conv_pack_items(items):
    """items: [(W (Cs,k,Ds) bf16 tap-major, out flat bf16, transpose)], up to 48 of any shapes, one launch (ttsk_win_conv_pack_items)."""
    n = len(items)
    arr = (L.PackItem * n)()
    for i, (W, out, tr) in enumerate(items):
        _dev(W, out)
        arr[i].src, arr[i].dst = W.data_ptr(), out.data_ptr()
        arr[i].Cs, arr[i].K, arr[i].Ds, arr[i].transpose = W.shape[0], W.shape[1], W.shape[2], int(tr)
    check(L.load().ttsk_win_conv_pack_items(arr, n, _stream()), "ttsk_win_conv_pack_items")


def win_conv_pack_table(items, device):
    """A device-resident item table for win_conv_pack_run: int32 tensor holding n `ttsk_pack_item`s (the tensors must stay where
    they are).  items as for win_conv_pack_items, any number."""
    n = len(items)
    arr = (L.PackItem * n)()
    for i, (W, out, tr) in enumerate(items):
        _dev(W, out)
        arr[i].src, arr[i].dst = W.data_ptr(), out.data_ptr()
        arr[i].Cs, arr[i].K, arr[i].Ds, arr[i].transpose = W.shape[0], W.shape[1], W.shape[2], int(tr)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.int32).clone()
    return host.to(device), n


def win_conv_pack_run(table, n):
    """Rewrite every pack of a win_conv_pack_table in one launch (ttsk_win_conv_pack_table)."""
    check(L.load().ttsk_win_conv_pack_table(_ptr(table), n, _stream()), "ttsk_win_conv_pack_table")


@_family("win_conv", _conv_flops)
def win_conv(x, packed, Cout, k, bias=None, relu=False, out_dtype=None, gate=None, delta_o32=None, delta_out=None):
    """[relu](Conv1d(Cin -> Cout, k)(x) + bias) on the window kernel, weights = a win_conv_pack_batch pack (ttsk_win_conv); `gate`
    (B,S,Cout) bf16: result zeroed where gate <= 0.  x (B,S,Cin) bf16 -> (B,S,Cout) bf16 or fp32."""
    _dev(x, packed, bias, gate)
    Bsz, S, Cin = x.shape
    out = torch.empty(Bsz, S, Cout, dtype=out_dtype or bf16, device=x.device)
    _dev(delta_o32, delta_out)
    check(L.load().ttsk_win_conv(_ptr(x), _ptr(packed), _ptr(bias), _ptr(gate), _ptr(delta_o32), _ptr(delta_out), _ptr(out),
                                 int(out.dtype == torch.float32), Bsz, S, Cin, Cout, k, int(relu), _stream()), "ttsk_win_conv")
    return out


@_family("win_conv", _conv_flops)
def win_conv_stats(x, packed, Cout, k, bias=None, frame_limit=None):
    """win_conv with fp32 output (Cin = 512) that also returns the BatchNorm statistics partials of its output
    ([B * ceil(S/64)][2*Cout] fp32: sum | sum of squares per tile) for bn_train(partials=...) (ttsk_win_conv_stats)."""
    _dev(x, packed, bias)
    Bsz, S, Cin = x.shape
    lib = L.load()
    out = torch.empty(Bsz, S, Cout, dtype=torch.float32, device=x.device)
    stats = _f32(lib.ttsk_win_conv_stats_rows(Bsz, S), 2 * Cout, device=x.device)
    lp, _ = _lim(frame_limit)
    check(lib.ttsk_win_conv_stats(_ptr(x), _ptr(packed), _ptr(bias), _ptr(out), _ptr(stats), lp, Bsz, S, Cin, Cout, k, _stream()),
          "ttsk_win_conv_stats")
    return out, stats


@_family("win_conv", _conv_flops)
def win_conv_bnb(x, packed, Cout, k, bn_x, mean, rstd, gamma, beta, use_tanh, p=0.0, keep=None, frame_limit=None):
    """win_conv with bf16 output (Cin = 512: a PostNet conv's input gradient on its transposed pack) that also returns the BatchNorm-
    backward statistics partials of the layer below, whose upstream gradient the output is ([B * ceil(S/64)][2*Cout] fp32: sum of dy | sum
    of dy * xhat per tile) for bn_bwd(partials=...) (ttsk_win_conv_bnb).  bn_x (B*S, Cout) fp32: that layer's conv output; keep: its
    bn_train keep bits."""
    _dev(x, packed, bn_x, mean, rstd, gamma, beta, keep)
    Bsz, S, Cin = x.shape
    lib = L.load()
    out = torch.empty(Bsz, S, Cout, dtype=bf16, device=x.device)
    stats = _f32(lib.ttsk_win_conv_stats_rows(Bsz, S), 2 * Cout, device=x.device)
    lp, _ = _lim(frame_limit)
    check(lib.ttsk_win_conv_bnb(_ptr(x), _ptr(packed), _ptr(out), _ptr(stats), _ptr(bn_x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta),
                                _ptr(keep), float(p), int(use_tanh), lp, Bsz, S, Cin, Cout, k, _stream()), "ttsk_win_conv_bnb")
    return out, stats


@_family("win_conv", lambda x, packed, resid, Cout, k: 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * Cout * k)
def win_conv_resid(x, packed, resid, Cout, k):
    """bf16 (B,S,Cout) = Conv1d(Cin -> Cout, k)(x) + resid (fp32 (B,S,Cout)) on the window kernel (ttsk_win_conv_resid): the PostNet's
    first conv's input gradient (512 -> 80 on the transposed pack) + the mel terms' own gradient."""
    _dev(x, packed, resid)
    Bsz, S, Cin = x.shape
    out = torch.empty(Bsz, S, Cout, dtype=bf16, device=x.device)
    check(L.load().ttsk_win_conv_resid(_ptr(x), _ptr(packed), _ptr(resid), _ptr(out), Bsz, S, Cin, Cout, k, _stream()), "ttsk_win_conv_resid")
    return out


@_family("win_conv", _conv_flops)
def win_conv_dual(x, packed, Cout, k, bias=None):
    """fp32 (B,S,Cout) = Conv1d(Cin -> Cout, k)(x) + bias on the window kernel, and its bf16 copy (ttsk_win_conv_dual: mel_linear)."""
    _dev(x, packed, bias)
    Bsz, S, Cin = x.shape
    out = torch.empty(Bsz, S, Cout, dtype=torch.float32, device=x.device)
    out16 = torch.empty(Bsz, S, Cout, dtype=bf16, device=x.device)
    check(L.load().ttsk_win_conv_dual(_ptr(x), _ptr(packed), _ptr(bias), _ptr(out), _ptr(out16), Bsz, S, Cin, Cout, k, _stream()), "ttsk_win_conv_dual")
    return out, out16


def win_pack_numel(Cs, k, Ds, transpose):
    """Elements of the fragment-major pack of a (Cs, k, Ds) tap-major weight (win_conv_pack_*): the contraction padded to whole 32-channel k-steps."""
    cout, cin = (Ds, Cs) if transpose else (Cs, Ds)
    return k * ((cin + 31) // 32) * 32 * cout


@_family("win_conv", _conv_flops)
def win_conv_split(x, packed, Cout, k):
    """An input-gradient conv with a wide contraction (x (B,S,n*256) bf16) as n window convs over 256-channel slices in one launch:
    fp32 Slabs (n, B*S*Cout) for layernorm_bwd(slabs=...) (ttsk_win_conv_split).  `packed`: the whole transposed pack."""
    _dev(x, packed)
    Bsz, S, Cin = x.shape
    n = Cin // 256
    ws = _f32(n, Bsz * S * Cout, device=x.device)
    check(L.load().ttsk_win_conv_split(_ptr(x), _ptr(packed), _ptr(ws), n, Bsz, S, Cin, Cout, k, _stream()), "ttsk_win_conv_split")
    return Slabs(ws, n, Bsz * S * Cout)


@_family("win_conv", lambda x, W, *a, **kw: 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * W.shape[0] * W.shape[1])
def ffn_conv_fwd(x, W, bias, relu=True, packed=None):
    """relu(Conv1d(256 -> Cout, k)(x) + bias) on the window kernel (ttsk_ffn_conv_fwd; SubLayers.py:93-101, w_1).
    x (B,S,256) bf16, W (Cout,k,256) bf16 tap-major -> (B,S,Cout) bf16; `packed`: ffn_pack_weight(W), read instead of W."""
    _dev(x, W, bias, packed)
    Bsz, S, Cin = x.shape
    Cout, k, _ = W.shape
    out = torch.empty(Bsz, S, Cout, dtype=bf16, device=x.device)
    check(L.load().ttsk_ffn_conv_fwd(_ptr(x), _ptr(W if packed is None else packed), _ptr(bias), _ptr(out), Bsz, S, Cin, Cout, k, int(relu),
                                     int(packed is not None), _stream()), "ttsk_ffn_conv_fwd")
    return out


def conv1d_dx(dy, W, dilation=1, out=None, R=None, G=None, **kw):
    """dx (B,T,Cin) = sum_j dy[t + pad - j*dil] @ W[:, j, :]."""
    Bsz, T, Cout = dy.shape
    _, k, Cin = W.shape
    pad = dilation * (k - 1) // 2
    if kw.get("raw"):
        return gemm(dy, W, None, Bsz * T, Cin, Cout, Cout, k * Cin, Cin, flags=B_TR, taps=k, seg_len=T, tap_shift0=pad,
                    tap_dshift=-dilation, b_tap_stride=Cin, **kw)
    if out is None:
        out = torch.empty(Bsz, T, Cin, dtype=bf16, device=dy.device)
    gemm(dy, W, out, Bsz * T, Cin, Cout, Cout, k * Cin, Cin, flags=B_TR, R=R, ldr=Cin, G=G, ldg=Cin, taps=k, seg_len=T,
         tap_shift0=pad, tap_dshift=-dilation, b_tap_stride=Cin, **kw)
    return out


def conv1d_dw(dy, x, dst, dilation=1, k=1, accumulate=True, **kw):
    """dst (Cout,k,Cin) fp32 (+)= sum_rows dy[r, co] * x[r + j*dil - pad, ci]  (taps as the second batch index)."""
    Bsz, T, Cout = dy.shape
    Cin = x.shape[2]
    pad = dilation * (k - 1) // 2
    rows = Bsz * T
    gemm(dy, x, dst, Cout, Cin, rows, Cout, Cin, k * Cin, flags=A_TR | B_TR | (ACCUM_C if accumulate else 0), nz2=k,
         sC=(0, Cin), bseg_len=T, bshift0=-pad, bdshift=dilation, **kw)
    return dst


def dwconv_supported(Cout, Cin, k):
    return bool(L.load().ttsk_dwconv_supported(int(Cout), int(Cin), int(k)))


def dwconv_batch(items):
    """Weight gradients of Conv1d layers with taps, one launch (ttsk_dwconv_batch, csrc/dwconv.hip).  items: [(dy (B,S,Cout) bf16,
    x (B,S,Cin) bf16, dst (Cout,k,Cin) fp32, lens int64 (B,) or None, accumulate)], all of one k, at most 12."""
    arr = (L.DwConvItem * len(items))()
    for i, (dy, x, dst, lens, accumulate) in enumerate(items):
        _dev(dy, x, dst, lens)
        Bsz, S, Cout = dy.shape
        Cin = x.shape[2]
        if dst.shape[0] != Cout or dst.shape[2] != Cin or not dst.is_contiguous() or dst.dtype != torch.float32:
            raise L.TtskError("dwconv_batch: dst must be a contiguous fp32 (Cout, k, Cin) tensor")
        if dy.stride(2) != 1 or x.stride(2) != 1 or dy.stride(0) != S * dy.stride(1) or x.stride(0) != S * x.stride(1) or dy.dtype != bf16 or x.dtype != bf16:
            raise L.TtskError("dwconv_batch: dy / x must be bf16 (B, S, C) with unit channel stride and utterances back to back")
        it = arr[i]
        it.dy, it.x, it.dw, it.lens = dy.data_ptr(), x.data_ptr(), dst.data_ptr(), _ptr(lens)
        it.Cout, it.Cin, it.K, it.ldy, it.ldx, it.B, it.S, it.accumulate = Cout, Cin, dst.shape[1], dy.stride(1), x.stride(1), Bsz, S, int(bool(accumulate))
    if GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.load().ttsk_dwconv_batch(arr, len(items), _stream()), "ttsk_dwconv_batch")
    if GEMM_TRACE is not None:
        e1.record()
        # algorithmic FLOPs: every row of every utterance (2*B*S*Cout*Cin*k), whatever `lens` lets the kernel skip
        fl = sum(2.0 * it[0].shape[0] * it[0].shape[1] * it[0].shape[2] * it[1].shape[2] * it[2].shape[1] for it in items)
        nwg = sum((it[0].shape[2] // 256) * (it[1].shape[2] // 32) for it in items)
        GEMM_TRACE.append((e0, e1, fl, "dwconv%d" % items[0][2].shape[1], (len(items), 0, 0, 1, 1, nwg)))
    if LAUNCH_COUNTS is not None:
        LAUNCH_COUNTS["dwconv"] = LAUNCH_COUNTS.get("dwconv", 0) + 1


def dwgemm_supported(Cout, Cin, k):
    return bool(L.load().ttsk_dwgemm_supported(int(Cout), int(Cin), int(k)))


def dwgemm_batch(items, max_wgs=0):
    """Weight gradients with Cout, Cin multiples of 256 on the 256x256-tile kernel (ttsk_dwgemm_batch, csrc/dwgemm.hip), one launch per
    28 problems.  items: [(dy (B,S,Cout) bf16, x (B,S,Cin) bf16, dst (Cout,k,Cin) fp32 or (Cout,Cin), lens int64 (B,) or None, accumulate,
    splits)].  Returns [(ReduceItem, slabs)] for the problems with splits > 1: run gemm_reduce_batch on them (DeferQueue does)."""
    lib = L.load()
    reduce = []
    for base in range(0, len(items), 28):
        chunk = items[base:base + 28]
        arr = (L.DwGemmItem * len(chunk))()
        keep = []
        for i, (dy, x, dst, lens, accumulate, splits) in enumerate(chunk):
            _dev(dy, x, dst, lens)
            Bsz, S, Cout = dy.shape
            Cin = x.shape[2]
            k = dst.shape[1] if dst.dim() == 3 else 1
            if dst.numel() != Cout * k * Cin or not dst.is_contiguous() or dst.dtype != torch.float32:
                raise L.TtskError("dwgemm_batch: dst must be a contiguous fp32 (Cout, k, Cin) tensor")
            if dy.stride(2) != 1 or x.stride(2) != 1 or dy.stride(0) != S * dy.stride(1) or x.stride(0) != S * x.stride(1) or dy.dtype != bf16 or x.dtype != bf16:
                raise L.TtskError("dwgemm_batch: dy / x must be bf16 (B, S, C) with unit channel stride and utterances back to back")
            splits = max(1, min(int(splits), Bsz))
            ws = torch.empty(splits * k * Cout * Cin, dtype=torch.float32, device=dy.device) if splits > 1 else None
            it = arr[i]
            it.dy, it.x, it.dw, it.workspace, it.lens = dy.data_ptr(), x.data_ptr(), dst.data_ptr(), _ptr(ws), _ptr(lens)
            it.Cout, it.Cin, it.K, it.ldy, it.ldx, it.B, it.S = Cout, Cin, k, dy.stride(1), x.stride(1), Bsz, S
            it.accumulate, it.splits = int(bool(accumulate)), splits
            if ws is not None:
                r = L.ReduceItem()
                r.ws, r.C, r.M, r.N, r.ldc, r.nz, r.splits = ws.data_ptr(), dst.data_ptr(), Cout, Cin, k * Cin, k, splits
                r.accumulate, r.sC2, r.alpha = int(bool(accumulate)), Cin, 1.0
                reduce.append((r, ws))
        if GEMM_TRACE is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(lib.ttsk_dwgemm_batch(arr, len(chunk), int(max_wgs), _stream()), "ttsk_dwgemm_batch")
        if GEMM_TRACE is not None:
            e1.record()
            fl = sum(2.0 * it[0].shape[0] * it[0].shape[1] * it[0].shape[2] * it[1].shape[2] * (it[2].shape[1] if it[2].dim() == 3 else 1) for it in chunk)
            GEMM_TRACE.append((e0, e1, fl, "dwgemm", (len(chunk), 0, 0, 1, 1, 0)))
        if LAUNCH_COUNTS is not None:
            LAUNCH_COUNTS["dwgemm"] = LAUNCH_COUNTS.get("dwgemm", 0) + 1
    return reduce


def conv_transpose1d(x, Wp, bias, stride, k, out=None, in_slope=0.0, flags=0, C2=None, out_slope=0.0, **kw):
    """ConvTranspose1d(padding=(k-stride)//2) as `stride` polyphase implicit GEMMs.
    x (B,T,Cin) bf16, Wp (k, Cout, Cin) bf16 (tap-major repack of torch's (Cin,Cout,k)) -> (B,T*stride,Cout).
    reference: hifi/models.py:166-176,189."""
    Bsz, T, Cin = x.shape
    Cout = Wp.shape[1]
    p = (k - stride) // 2
    if out is None:
        out = torch.empty(Bsz, T * stride, Cout, dtype=x.dtype, device=x.device)
    taps = k // stride
    # phases r = 0..stride-1; phases with the same first tap offset qoff share one launch (batch index z2 = phase):
    # their weights Wp[r] are consecutive, their output rows differ by one (out_add_dz = 1)
    groups = []
    for r in range(stride):
        qoff = max(0, -((r - p) // stride))          # ceil((p - r)/stride) clipped at 0
        if groups and groups[-1][1] == qoff:
            groups[-1][2] += 1
        else:
            groups.append([r, qoff, 1])
    for r0, qoff, n in groups:
        gemm(x, Wp[r0], out, Bsz * T, Cout, Cin, Cin, Cin, Cout, flags=flags | (LRELU_IN if in_slope else 0),
             bias=bias, taps=taps, seg_len=T, tap_shift0=qoff, tap_dshift=-1, b_tap_stride=stride * Cout * Cin,
             out_seg=T * stride, out_mul=stride, out_add=qoff * stride + r0 - p, out_add_dz=1, nz2=n, sB=(0, Cout * Cin),
             in_slope=in_slope, C2=C2, out_slope=out_slope, **kw)
    return out


# ----------------------------------------------------------------------------------------------------------------
# row kernels

def _f32(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


def layernorm_fwd(y, res, gamma, beta, lens=None, seg_len=0, p_pre=0.0, site_pre=0, p_post=0.0, site_post=0, rng=None,
                  save_z=True, head=None, want_out=True, eps=1e-5, out=None):
    """Fused block tail (see include/ttsk.h).  y/res (rows,D) bf16.  Returns (out, z, mean, rstd, head_out)."""
    _dev(y, res, gamma, beta, lens, rng)
    rows, D = y.shape
    dev = y.device
    if out is None:
        out = torch.empty(rows, D, dtype=bf16, device=dev) if want_out else None
    z = torch.empty(rows, D, dtype=bf16, device=dev) if save_z else None
    mean, rstd = _f32(rows, device=dev), _f32(rows, device=dev)
    hw = hb = ho = None
    if head is not None:
        hw, hb = head
        ho = _f32(rows, device=dev)
    check(L.load().ttsk_layernorm_fwd(_ptr(y), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(z), _ptr(mean), _ptr(rstd),
                                      _ptr(lens), seg_len, rows, D, eps, p_pre, site_pre, p_post, site_post, _ptr(rng),
                                      _ptr(hw), _ptr(hb), _ptr(ho), _stream()), "ttsk_layernorm_fwd")
    return out, z, mean, rstd, ho


def gemm_ln_fwd(x, W, bias, res, gamma, beta, lens=None, seg_len=0, p_pre=0.0, site_pre=0, rng=None, save_z=True, eps=1e-5, out=None):
    """LayerNorm(dropout(x @ W^T + bias) + res) with PAD rows zeroed, one kernel (D = 256).  x (rows, K) bf16, W (256, K) or
    (256, 1, K) bf16.  Returns (out, z, mean, rstd) like layernorm_fwd."""
    _dev(x, W, bias, res, gamma, beta, lens, rng)
    rows, K = x.shape
    D = W.shape[0]
    W2 = W.reshape(D, -1)
    dev = x.device
    if out is None:
        out = torch.empty(rows, D, dtype=bf16, device=dev)
    z = torch.empty(rows, D, dtype=bf16, device=dev) if save_z else None
    mean, rstd = _f32(rows, device=dev), _f32(rows, device=dev)
    check(L.load().ttsk_gemm_ln_fwd(_ptr(x), x.stride(0), _ptr(W2), W2.stride(0), _ptr(bias), _ptr(res), _ptr(gamma), _ptr(beta),
                                    _ptr(out), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(lens), seg_len, rows, K, D, eps, p_pre, site_pre,
                                    _ptr(rng), _stream()), "ttsk_gemm_ln_fwd")
    return out, z, mean, rstd


def win_ln_supported(K, D):
    return bool(L.load().ttsk_win_ln_supported(K, D))


@_family("win_ln", lambda x, *a, **kw: 2.0 * x.shape[0] * x.shape[1] * 256 + (2.0 * x.shape[0] * 256 * 768 if kw.get("proj") is not None else 0.0))
def win_ln_fwd(x, packed, bias, res, gamma, beta, lens=None, seg_len=0, p_pre=0.0, site_pre=0, rng=None, save_z=True, eps=1e-5, out=None,
               proj=None):
    """gemm_ln_fwd with the weight as a fragment-major pack (win_conv_pack_*; ttsk_win_ln_fwd): x (rows, K) bf16, K = 256 or 1024.
    proj = (packed q|k|v weight, bias fp32[768]): the next block's q|k|v projection of the output rows in the same launch
    (ttsk_win_ln_proj_fwd); then returns (out, z, mean, rstd, qkv)."""
    _dev(x, packed, bias, res, gamma, beta, lens, rng)
    rows, K = x.shape
    D = 256
    dev = x.device
    if out is None:
        out = torch.empty(rows, D, dtype=bf16, device=dev)
    z = torch.empty(rows, D, dtype=bf16, device=dev) if save_z else None
    mean, rstd = _f32(rows, device=dev), _f32(rows, device=dev)
    if proj is not None:
        pw, pb = proj
        _dev(pw, pb)
        qkv = torch.empty(rows, 3 * D, dtype=bf16, device=dev)
        check(L.load().ttsk_win_ln_proj_fwd(_ptr(x), x.stride(0), _ptr(packed), _ptr(bias), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(z),
                                            _ptr(mean), _ptr(rstd), _ptr(lens), seg_len, rows, K, D, eps, p_pre, site_pre, _ptr(rng), _ptr(pw),
                                            _ptr(pb), 3 * D, _ptr(qkv), _stream()), "ttsk_win_ln_proj_fwd")
        return out, z, mean, rstd, qkv
    check(L.load().ttsk_win_ln_fwd(_ptr(x), x.stride(0), _ptr(packed), _ptr(bias), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(z),
                                   _ptr(mean), _ptr(rstd), _ptr(lens), seg_len, rows, K, D, eps, p_pre, site_pre, _ptr(rng), _stream()),
          "ttsk_win_ln_fwd")
    return out, z, mean, rstd


def layernorm_bwd(dout, z, mean, rstd, gamma, beta, lens=None, seg_len=0, relu_in=False, p_pre=0.0, site_pre=0,
                  p_post=0.0, site_post=0, rng=None, dhead=None, head_w=None, want_dz=True, slabs=None, R=None):
    """Returns (dz, dy, partials, nblk).  dy is dz when p_pre == 0.  partials layout: see include/ttsk.h.
    `slabs` (a Slabs from gemm(raw=True)) + `R` (bf16 residual) replace `dout`: dout = sum of the slabs + R, summed in fp32
    while the rows are read (ttsk_layernorm_bwd_slabs)."""
    _dev(dout, z, dhead, R)
    rows, D = z.shape
    dev = z.device
    lib = L.load()
    nblk = lib.ttsk_layernorm_bwd_nblocks(rows)
    ncol = (4 * D + 1) if dhead is not None else 3 * D
    partials = _f32(nblk, ncol, device=dev)
    dz = torch.empty(rows, D, dtype=bf16, device=dev) if (want_dz or p_pre == 0.0) else None
    dy = torch.empty(rows, D, dtype=bf16, device=dev) if p_pre > 0.0 else None
    if slabs is not None:
        check(lib.ttsk_layernorm_bwd_slabs(_ptr(slabs.ws), slabs.splits, slabs.stride, _ptr(R), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                           _ptr(beta), _ptr(lens), seg_len, rows, D, int(relu_in), p_pre, site_pre, p_post, site_post,
                                           _ptr(rng), _ptr(dz), _ptr(dy), _ptr(partials), _stream()), "ttsk_layernorm_bwd_slabs")
        return dz, (dy if dy is not None else dz), partials, nblk
    check(lib.ttsk_layernorm_bwd(_ptr(dout), _ptr(dhead), _ptr(head_w), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                 _ptr(beta), _ptr(lens), seg_len, rows, D, int(relu_in), p_pre, site_pre, p_post,
                                 site_post, _ptr(rng), _ptr(dz), _ptr(dy), _ptr(partials), _stream()), "ttsk_layernorm_bwd")
    return dz, (dy if dy is not None else dz), partials, nblk


@_family("ln_bwd_proj", lambda dout, z, mean, rstd, gamma, packed, Cout, *a, **kw: 2.0 * z.shape[0] * 256 * Cout + (2.0 * z.shape[0] * 768 * 256 if kw.get("pre") is not None else 0.0))
def layernorm_bwd_proj(dout, z, mean, rstd, gamma, packed, Cout, lens=None, seg_len=0, p_pre=0.0, site_pre=0, rng=None, slabs=None, R=None,
                       gate=None, delta_o32=None, delta_out=None, pre=None):
    """layernorm_bwd (D = 256) and the k = 1 window conv on its dy in ONE launch (ttsk_layernorm_bwd_proj): `packed` is the
    win_conv pack of the transposed weight, Cout 256 or 1024; gate / delta as for win_conv.
    pre = (x (rows, 768) bf16, packed transposed weight): the upstream gradient is x · W' (+ R), computed inside the kernel too
    (instead of dout / slabs).  Returns (dz, dy, partials, nblk, out)."""
    _dev(dout, z, R, gate, delta_o32, delta_out, packed)
    px, pw = pre if pre is not None else (None, None)
    _dev(px, pw)
    rows, D = z.shape
    dev = z.device
    lib = L.load()
    nblk = lib.ttsk_layernorm_bwd_proj_nblocks(rows)
    partials = _f32(nblk, 3 * D, device=dev)
    dz = torch.empty(rows, D, dtype=bf16, device=dev)
    dy = torch.empty(rows, D, dtype=bf16, device=dev) if p_pre > 0.0 else None
    out = torch.empty(rows, Cout, dtype=bf16, device=dev)
    check(lib.ttsk_layernorm_bwd_proj(_ptr(dout), _ptr(slabs.ws) if slabs is not None else None, slabs.splits if slabs is not None else 0,
                                      slabs.stride if slabs is not None else 0, _ptr(R), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                      _ptr(lens), seg_len, rows, D, p_pre, site_pre, _ptr(rng), _ptr(dz), _ptr(dy), _ptr(partials),
                                      _ptr(packed), Cout, _ptr(gate), _ptr(delta_o32), _ptr(delta_out), _ptr(out), _ptr(px), _ptr(pw),
                                      px.shape[1] if px is not None else 0, _stream()),
          "ttsk_layernorm_bwd_proj")
    return dz, (dy if dy is not None else dz), partials, nblk, out


@_family("ln_bwd_proj", lambda dqkv, *a, **kw: 2.0 * dqkv.shape[0] * dqkv.shape[1] * 256)
def qkv_dx(dqkv, packed, R=None):
    """dqkv (rows, 768) bf16 x the packed transposed q|k|v weight (+ R) -> (rows, 256) bf16 (ttsk_qkv_dx)."""
    _dev(dqkv, packed, R)
    rows, K = dqkv.shape
    out = torch.empty(rows, 256, dtype=bf16, device=dqkv.device)
    check(L.load().ttsk_qkv_dx(_ptr(dqkv), _ptr(packed), _ptr(R), _ptr(out), rows, K, 256, _stream()), "ttsk_qkv_dx")
    return out


def layernorm_fwd_grouped(y, gamma, beta, groups, param_stride, site_stride, lens=None, seg_len=0, p_post=0.0, site_post=0, rng=None,
                          head=None, want_out=True, eps=1e-5):
    """`groups` independent LayerNorm tails in one launch (the three VariancePredictors): y (groups*group_rows, D) bf16; group g
    takes its gamma / beta / head at + g*param_stride floats and dropout site site_post + g*site_stride.  -> (out, mean, rstd, head_out)."""
    _dev(y, gamma, beta, lens, rng)
    rows, D = y.shape
    dev = y.device
    out = torch.empty(rows, D, dtype=bf16, device=dev) if want_out else None
    mean, rstd = _f32(rows, device=dev), _f32(rows, device=dev)
    hw = hb = ho = None
    if head is not None:
        hw, hb = head
        ho = _f32(rows, device=dev)
    check(L.load().ttsk_layernorm_fwd_grouped(_ptr(y), None, _ptr(gamma), _ptr(beta), _ptr(out), None, _ptr(mean), _ptr(rstd), _ptr(lens),
                                              seg_len, groups, rows // groups, param_stride, site_stride, D, eps, 0.0, 0, p_post, site_post,
                                              _ptr(rng), _ptr(hw), _ptr(hb), _ptr(ho), _stream()), "ttsk_layernorm_fwd_grouped")
    return out, mean, rstd, ho


def layernorm_bwd_grouped(dout, z, mean, rstd, gamma, beta, groups, param_stride, site_stride, lens=None, seg_len=0, relu_in=False,
                          p_post=0.0, site_post=0, rng=None, dhead=None, head_w=None):
    """Backward of layernorm_fwd_grouped.  Returns (dz, partials [groups][nblk][ncol], nblk per group)."""
    _dev(dout, z, dhead)
    rows, D = z.shape
    dev = z.device
    lib = L.load()
    nblk = lib.ttsk_layernorm_bwd_nblocks(rows // groups)
    ncol = (4 * D + 1) if dhead is not None else 3 * D
    partials = _f32(groups, nblk, ncol, device=dev)
    dz = torch.empty(rows, D, dtype=bf16, device=dev)
    check(lib.ttsk_layernorm_bwd_grouped(_ptr(dout), _ptr(dhead), _ptr(head_w), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta),
                                         _ptr(lens), seg_len, groups, rows // groups, param_stride, site_stride, D, int(relu_in), 0.0, 0,
                                         p_post, site_post, _ptr(rng), _ptr(dz), None, _ptr(partials), _stream()), "ttsk_layernorm_bwd_grouped")
    return dz, partials, nblk


def va_embed(stack, speakers, spk_table, Lp, pitch_t, pitch_bins, pitch_table, energy_t, energy_bins, energy_table, row_limit=None):
    """stack (3, rows, D) bf16 with stack[0] = encoder output: fills stack[1] = x + speaker, stack[2] = stack[1] + pitch embedding and
    returns (x3 = stack[2] + energy embedding, pitch bucket indices, energy bucket indices).  reference: modules.py:158-193."""
    _dev(stack, speakers, spk_table, pitch_t, pitch_bins, pitch_table, energy_t, energy_bins, energy_table)
    _, rows, D = stack.shape
    x3 = torch.empty(rows, D, dtype=bf16, device=stack.device)
    pidx = torch.empty(rows, dtype=torch.int32, device=stack.device)
    eidx = torch.empty(rows, dtype=torch.int32, device=stack.device)
    check(L.load().ttsk_va_embed(_ptr(stack[0]), _ptr(spk_table), _ptr(speakers), Lp, _ptr(pitch_t), _ptr(pitch_bins), _ptr(pitch_table),
                                 _ptr(energy_t), _ptr(energy_bins), _ptr(energy_table), pitch_bins.numel(), _ptr(stack[1]), _ptr(stack[2]),
                                 _ptr(x3), _ptr(pidx), _ptr(eidx), rows, D, _ptr(row_limit), _stream()), "ttsk_va_embed")
    return x3, pidx, eidx


def va_combine(dx3, dxin, Lp=0, row_limit=None):
    """dx3 (rows, D) bf16, dxin (3, rows, D) fp32 -> (dx2, dx1, dx) bf16: see include/ttsk.h."""
    _dev(dx3, dxin, row_limit)
    rows, D = dx3.shape
    dx2, dx1, dx = torch.empty_like(dx3), torch.empty_like(dx3), torch.empty_like(dx3)
    check(L.load().ttsk_va_combine(_ptr(dx3), _ptr(dxin), _ptr(dx2), _ptr(dx1), _ptr(dx), rows, D, Lp, _ptr(row_limit), _stream()), "ttsk_va_combine")
    return dx2, dx1, dx


def colsum_finalize(partials, nblk, ncols, ld, dst, accumulate=True, scale=1.0, defer=None):
    """dst[c] (+)= scale * sum_b partials[b*ld + c].  `defer`: a list — the item is queued for flush_finalize instead."""
    if defer is not None:
        it = L.FinalizeItem()
        it.partials, it.dst, it.nblk, it.ncols, it.ld, it.accumulate, it.scale = partials.data_ptr(), dst.data_ptr(), nblk, ncols, ld, int(accumulate), scale
        defer.append((it, partials))
        return dst
    check(L.load().ttsk_colsum_finalize(_ptr(partials), nblk, ncols, ld, _ptr(dst), int(accumulate), scale, _stream()),
          "ttsk_colsum_finalize")
    return dst


def flush_finalize(items):
    """One launch each for the queued embedding scatter-sums, the queued column sums (bias gradients) and then all queued
    finalisations (ttsk_scatter_sum_batch, ttsk_colsum_batch, ttsk_colsum_finalize_batch)."""
    if not items:
        return
    if LAUNCH_COUNTS is not None:
        LAUNCH_COUNTS["colsum_flush"] = LAUNCH_COUNTS.get("colsum_flush", 0) + 1
    sc = [it for it, _ in items if isinstance(it, L.ScatterItem)]
    if sc:
        check(L.load().ttsk_scatter_sum_batch((L.ScatterItem * len(sc))(*sc), len(sc), _stream()), "ttsk_scatter_sum_batch")
    cs = [it for it, _ in items if isinstance(it, L.ColsumItem)]
    if cs:
        check(L.load().ttsk_colsum_batch((L.ColsumItem * len(cs))(*cs), len(cs), _stream()), "ttsk_colsum_batch")
    fin = [(it, keep) for it, keep in items if isinstance(it, L.FinalizeItem)]
    items[:] = fin
    if items:
        arr = (L.FinalizeItem * len(items))(*[it for it, _ in items])
        check(L.load().ttsk_colsum_finalize_batch(arr, len(items), _stream()), "ttsk_colsum_finalize_batch")
    items.clear()


def colsum_into(x, dst, accumulate=True, defer=None):
    """dst[C] (+)= column sums of x (rows, C) — bias gradients."""
    _dev(x, dst)
    rows, Cn = x.shape
    lib = L.load()
    nblk = lib.ttsk_colsum_nblocks(rows)
    partials = _f32(nblk, Cn, device=x.device)
    if defer is not None:          # queue the column sum itself as well; `x` stays alive until flush_finalize
        it = L.ColsumItem()
        it.x, it.partials, it.is_f32, it.rows, it.C, it.ld, it.nblk = x.data_ptr(), partials.data_ptr(), int(x.dtype == torch.float32), rows, Cn, x.stride(0), nblk
        defer.append((it, x))
    else:
        check(lib.ttsk_colsum(_ptr(x), int(x.dtype == torch.float32), rows, Cn, x.stride(0), _ptr(partials), _stream()), "ttsk_colsum")
    return colsum_finalize(partials, nblk, Cn, Cn, dst, accumulate, defer=defer)


@_family("flash_attention", lambda qkv, lens, Bn, H, S, *a, **kw: 4.0 * Bn * S * S * (qkv.shape[1] // 3))
def flash_attention_fwd(qkv, lens, Bn, H, S, want_lse):
    """softmax(Q K^T / sqrt(128) + key mask) V on the q|k|v buffer (rows, 3d) -> (o (rows, d) bf16, lse (B*H, S) fp32 or None,
    o32 (rows, d) fp32 or None) with no S x S tensor in HBM (ttsk_flash_attention_fwd); lse and o32 are what the backward needs."""
    _dev(qkv, lens)
    d = qkv.shape[1] // 3
    o = torch.empty(Bn * S, d, dtype=bf16, device=qkv.device)
    lse = _f32(Bn * H, S, device=qkv.device) if want_lse else None
    o32 = _f32(Bn * S, d, device=qkv.device) if want_lse else None
    check(L.load().ttsk_flash_attention_fwd(_ptr(qkv), _ptr(o), _ptr(o32), _ptr(lse), _ptr(lens), Bn, H, S, d, (d // H) ** -0.5, _stream()),
          "ttsk_flash_attention_fwd")
    return o, lse, o32


@_family("flash_attention", lambda qkv, o, do, lse, lens, Bn, H, S, *a, **kw: 8.0 * Bn * S * S * (qkv.shape[1] // 3))
def flash_attention_bwd(qkv, o, do, lse, lens, Bn, H, S, o32=None, delta=None):
    """dqkv (rows, 3d) bf16 = gradients of q | k | v (ttsk_flash_attention_bwd: delta, then dQ and dK/dV sides as one grid; P
    recomputed from lse).  `delta` (B*H, S) fp32: already computed by the producer of `do` (win_conv(delta_out=...)): no delta launch."""
    _dev(qkv, o, do, lse, lens, o32, delta)
    d = qkv.shape[1] // 3
    dqkv = torch.empty(Bn * S, 3 * d, dtype=bf16, device=qkv.device)
    ready = delta is not None
    if delta is None:
        delta = _f32(Bn * H, S, device=qkv.device)
    check(L.load().ttsk_flash_attention_bwd(_ptr(qkv), _ptr(o), _ptr(o32), _ptr(do), _ptr(lse), _ptr(delta), int(ready), _ptr(dqkv), _ptr(lens),
                                            Bn, H, S, d, (d // H) ** -0.5, _stream()), "ttsk_flash_attention_bwd")
    return dqkv


def softmax_fwd(scores, lens, H):
    """scores (nz,S,Sp) fp32 -> probs (nz,S,Sp) bf16; keys >= lens[z // H] masked."""
    _dev(scores, lens)
    nz, S, Sp = scores.shape
    probs = torch.empty(nz, S, Sp, dtype=bf16, device=scores.device)
    check(L.load().ttsk_softmax_fwd(_ptr(scores), _ptr(probs), _ptr(lens), nz, H, S, Sp, _stream()), "ttsk_softmax_fwd")
    return probs


def softmax_bwd(probs, dprobs, alpha):
    _dev(probs, dprobs)
    nz, S, Sp = probs.shape
    ds = torch.empty(nz, S, Sp, dtype=bf16, device=probs.device)
    check(L.load().ttsk_softmax_bwd(_ptr(probs), _ptr(dprobs), _ptr(ds), nz, S, Sp, alpha, _stream()), "ttsk_softmax_bwd")
    return ds


def bucketize(values, bins, scale=1.0, want_scaled=False):
    _dev(values, bins)
    values = values.contiguous()
    idx = torch.empty(values.shape, dtype=torch.int32, device=values.device)
    scaled = torch.empty_like(values) if want_scaled else None
    check(L.load().ttsk_bucketize(_ptr(values), _ptr(bins), bins.numel(), scale, _ptr(idx), _ptr(scaled), values.numel(),
                                  _stream()), "ttsk_bucketize")
    return (idx, scaled) if want_scaled else idx


def duration_round(logd, d_control=1.0):
    _dev(logd)
    out = torch.empty_like(logd)
    check(L.load().ttsk_duration_round(_ptr(logd), d_control, _ptr(out), logd.numel(), _stream()), "ttsk_duration_round")
    return out


def length_mask(lens, T):
    """(B,) int64 -> (B,T) bool, True = PAD.  reference: fs_two/utils/tools.py:121-131."""
    _dev(lens)
    B = lens.shape[0]
    mask = torch.empty(B, T, dtype=torch.bool, device=lens.device)
    check(L.load().ttsk_length_mask(_ptr(lens), _ptr(mask), B, T, _stream()), "ttsk_length_mask")
    return mask


def add_f32(a, b, out=None, scale_b=1.0):
    """out = a + scale_b * b (fp32, same shapes)."""
    _dev(a, b)
    if out is None:
        out = torch.empty_like(a)
    check(L.load().ttsk_add_f32(_ptr(a), _ptr(b), scale_b, _ptr(out), a.numel(), _stream()), "ttsk_add_f32")
    return out


def gather_add(x, table, idx, idx_div=1, pe=None, pe_mod=1, rows=None, out=None):
    """out[row] = (x[row] if x is not None) + table[idx[row // idx_div]] + pe[row % pe_mod]; bf16 (rows, D)."""
    _dev(x, table, idx, pe)
    D = table.shape[1]
    if rows is None:
        rows = x.shape[0]
    if out is None:
        out = torch.empty(rows, D, dtype=bf16, device=table.device)
    check(L.load().ttsk_gather_add(_ptr(x), _ptr(table), _ptr(idx), int(idx.dtype == torch.int64), idx_div, _ptr(pe), pe_mod,
                                   _ptr(out), rows, D, _stream()), "ttsk_gather_add")
    return out


def scatter_sum(dx, idx, dtable, idx_div=1, skip_row=-1, accumulate=True, defer=None):
    """dtable[v] (+)= sum of the rows of dx whose index is v.  `defer` (the list flush_finalize drains): the embedding-table
    gradients of a backward pass are independent and nothing but the optimiser reads them — queued, they share one launch."""
    _dev(dx, idx, dtable)
    V, D = dtable.shape
    if defer is not None:
        it = L.ScatterItem()
        it.dx, it.idx, it.dtable = dx.data_ptr(), idx.data_ptr(), dtable.data_ptr()
        it.idx_is_i64, it.idx_div, it.n_idx, it.n_table_rows, it.D = int(idx.dtype == torch.int64), idx_div, idx.numel(), V, D
        it.skip_row, it.accumulate = skip_row, int(accumulate)
        defer.append((it, (dx, idx)))
        return dtable
    check(L.load().ttsk_scatter_sum(_ptr(dx), _ptr(idx), int(idx.dtype == torch.int64), idx_div, idx.numel(), _ptr(dtable), V, D,
                                    skip_row, int(accumulate), _stream()), "ttsk_scatter_sum")
    return dtable


def cast_bf16(src, dst=None):
    _dev(src)
    if dst is None:
        dst = torch.empty(src.shape, dtype=bf16, device=src.device)
    check(L.load().ttsk_cast_bf16(_ptr(src), _ptr(dst), src.numel(), _stream()), "ttsk_cast_bf16")
    return dst


def nct_to_ntc(x, dtype=bf16):
    """(B,C,T) fp32 contiguous -> (B,T,C) bf16 / fp16."""
    _dev(x)
    B, Cn, T = x.shape
    out = torch.empty(B, T, Cn, dtype=dtype, device=x.device)
    check(L.load().ttsk_nct_to_ntc(_ptr(x.contiguous()), _ptr(out), int(dtype == f16), B, Cn, T, _stream()), "ttsk_nct_to_ntc")
    return out


def to_int16(x, scale):
    _dev(x)
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    check(L.load().ttsk_to_int16(_ptr(x.contiguous()), _ptr(out), x.numel(), scale, _stream()), "ttsk_to_int16")
    return out


_PINNED = {}
_PINNED_LOCK = threading.Lock()


def to_host(t):
    """A device tensor on the host, through a pinned staging buffer kept per (device, shape, dtype) — at most eight of them: the pageable
    `.cpu()` of a waveform (0.2 MB for one utterance, 1.5 MB for a batch of eight) goes through the driver's own staging and takes 0.1-0.3 ms,
    a third to a seventh of that from pinned memory.  The caller gets its own copy (the staging buffer is reused by the next call).  The copy
    runs on the current stream of the TENSOR's device and that stream is the one waited for (not the current device's: `HIFIapi(device=
    "cuda:1")`); lookup, copy and clone happen under one lock, so two threads with same-shaped outputs cannot read each other's samples."""
    _dev(t)
    key = (t.device.index, tuple(t.shape), t.dtype)
    with _PINNED_LOCK:
        buf = _PINNED.get(key)
        if buf is None:
            if len(_PINNED) >= 8:
                _PINNED.pop(next(iter(_PINNED)))
            buf = _PINNED[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        with torch.cuda.device(t.device):
            buf.copy_(t, non_blocking=True)
            torch.cuda.current_stream(t.device).synchronize()
        return buf.clone()


# ---------------------------------------------------------------------------------------------------- batch norm

def _lim(frame_limit):
    """(pointer, seg_len) of a frame limit (None, or (int32[1] device tensor, frames per utterance in the row layout))."""
    return (None, 0) if frame_limit is None else (frame_limit[0].data_ptr(), int(frame_limit[1]))


def zero_frames_from(x, frame_limit):
    """x (rows, C) bf16/fp32, rows = utterances * seg_len: frames t >= frame_limit of every utterance := 0."""
    _dev(x)
    lp, seg = _lim(frame_limit)
    rows, Cn = x.shape
    check(L.load().ttsk_zero_frames_from(_ptr(x), x.element_size(), rows, Cn, seg, lp, _stream()), "ttsk_zero_frames_from")
    return x


BN_SLAB = True      # two-launch BatchNorm (channel slabs) where the channel count allows; the three-launch kernels serve the rest and inference


def bn_train_stats(x, running_mean=None, running_var=None, nbt=None, eps=1e-5, momentum=0.1, frame_limit=None):
    """x (rows, C) bf16 or fp32 -> (mean, rstd) fp32 of the batch; updates the running buffers in place."""
    _dev(x)
    rows, Cn = x.shape
    lib = L.load()
    nblk = lib.ttsk_bn_nblocks(rows)
    partials = _f32(nblk, 2 * Cn, device=x.device)
    mean, rstd = _f32(Cn, device=x.device), _f32(Cn, device=x.device)
    lp, seg = _lim(frame_limit)
    check(lib.ttsk_bn_stats(_ptr(x), int(x.dtype == torch.float32), rows, Cn, _ptr(partials), lp, seg, _stream()), "ttsk_bn_stats")
    check(lib.ttsk_bn_finalize(_ptr(partials), nblk, Cn, rows, eps, momentum, _ptr(mean), _ptr(rstd), _ptr(running_mean),
                               _ptr(running_var), _ptr(nbt), lp, seg, _stream()), "ttsk_bn_finalize")
    return mean, rstd


def rsqrt_eps(var, eps=1e-5):
    out = torch.empty_like(var)
    check(L.load().ttsk_rsqrt_eps(_ptr(var), eps, _ptr(out), var.numel(), _stream()), "ttsk_rsqrt_eps")
    return out


def bn_apply(x, mean, rstd, gamma, beta, use_tanh, p=0.0, site=0, rng=None, resid=None, out_f32=False, frame_limit=None):
    rows, Cn = x.shape
    o16 = None if out_f32 else torch.empty(rows, Cn, dtype=bf16, device=x.device)
    o32 = _f32(rows, Cn, device=x.device) if out_f32 else None
    check(L.load().ttsk_bn_apply(_ptr(x), int(x.dtype == torch.float32), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), rows, Cn, int(use_tanh), p, site,
                                 _ptr(rng), _ptr(resid), _ptr(o16), _ptr(o32), *_lim(frame_limit), _stream()), "ttsk_bn_apply")
    return o32 if out_f32 else o16


def bn_slab_supported(Cn):
    return BN_SLAB and Cn % 4 == 0 and (Cn % 64 == 0 or Cn <= 128)


@_family("batchnorm", lambda *a, **kw: 0.0)
def bn_train(x, running_mean, running_var, nbt, gamma, beta, use_tanh, p=0.0, site=0, rng=None, resid=None, out_f32=False,
             frame_limit=None, eps=1e-5, momentum=0.1, partials=None, want_keep=False):
    """Training-mode BatchNorm forward in two launches: batch statistics partials, then normalise (+tanh, dropout, residual)
    with mean / rstd finished per channel slab inside the second kernel.  -> (out, mean, rstd).  partials: rows already
    produced elsewhere ([nblk][2C] fp32) skip the first launch.  want_keep: -> (out, mean, rstd, keep), keep = the dropout's
    keep bits (uint8 [rows][C/4], None when p == 0) for bn_bwd(keep=...)."""
    _dev(x)
    rows, Cn = x.shape
    lib = L.load()
    lp, seg = _lim(frame_limit)
    xf = int(x.dtype == torch.float32)
    if partials is None:
        partials = _f32(lib.ttsk_bn_nchunks(rows), 2 * Cn, device=x.device)
        check(lib.ttsk_bn_stats_slab(_ptr(x), xf, rows, Cn, _ptr(partials), lp, seg, _stream()), "ttsk_bn_stats_slab")
    mean, rstd = _f32(Cn, device=x.device), _f32(Cn, device=x.device)
    o16 = None if out_f32 else torch.empty(rows, Cn, dtype=bf16, device=x.device)
    o32 = _f32(rows, Cn, device=x.device) if out_f32 else None
    keep = torch.empty(rows, Cn // 4, dtype=torch.uint8, device=x.device) if (want_keep and p > 0.0) else None
    check(lib.ttsk_bn_train_apply(_ptr(x), xf, _ptr(partials), partials.shape[0], eps, momentum, _ptr(mean), _ptr(rstd), _ptr(running_mean),
                                  _ptr(running_var), _ptr(nbt), _ptr(gamma), _ptr(beta), rows, Cn, int(use_tanh), p, site, _ptr(rng),
                                  _ptr(resid), _ptr(o16), _ptr(o32), _ptr(keep), lp, seg, _stream()), "ttsk_bn_train_apply")
    return ((o32 if out_f32 else o16), mean, rstd, keep) if want_keep else ((o32 if out_f32 else o16), mean, rstd)


@_family("batchnorm", lambda *a, **kw: 0.0)
def bn_bwd(dout, x, mean, rstd, gamma, beta, use_tanh, p=0.0, site=0, rng=None, dgamma=None, dbeta=None, frame_limit=None, keep=None,
           accumulate=True, partials=None):
    """dx (rows,C) bf16; dgamma/dbeta (fp32, accumulated in place when given, or overwritten with accumulate=False).  keep: bn_train's
    keep bits (slab kernels only).  partials: the statistics partial rows when the conv that produced `dout` emitted them (win_conv_bnb)."""
    rows, Cn = x.shape
    lib = L.load()
    if bn_slab_supported(Cn):
        f32, xf = int(dout.dtype == torch.float32), int(x.dtype == torch.float32)
        lp, seg = _lim(frame_limit)
        if partials is None:
            partials = _f32(lib.ttsk_bn_nchunks(rows), 2 * Cn, device=x.device)
            check(lib.ttsk_bn_bwd_stats_slab(_ptr(dout), f32, _ptr(x), xf, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), rows, Cn, int(use_tanh),
                                             p, site, _ptr(rng), _ptr(keep), _ptr(partials), lp, seg, _stream()), "ttsk_bn_bwd_stats_slab")
        dx = torch.empty(rows, Cn, dtype=bf16, device=x.device)
        check(lib.ttsk_bn_bwd_apply_slab(_ptr(dout), f32, _ptr(x), xf, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), rows, Cn, int(use_tanh),
                                         p, site, _ptr(rng), _ptr(keep), _ptr(partials), partials.shape[0], _ptr(dx), _ptr(dgamma), _ptr(dbeta), int(accumulate),
                                         lp, seg, _stream()), "ttsk_bn_bwd_apply_slab")
        return dx
    nblk = lib.ttsk_bn_nblocks(rows)
    partials = _f32(nblk, 2 * Cn, device=x.device)
    sums = _f32(2 * Cn, device=x.device)
    f32 = int(dout.dtype == torch.float32)
    check(lib.ttsk_bn_bwd_stats(_ptr(dout), f32, _ptr(x), int(x.dtype == torch.float32), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), rows, Cn,
                                int(use_tanh), p, site, _ptr(rng), _ptr(partials), *_lim(frame_limit), _stream()), "ttsk_bn_bwd_stats")
    colsum_finalize(partials, nblk, 2 * Cn, 2 * Cn, sums, accumulate=False)
    dx = torch.empty(rows, Cn, dtype=bf16, device=x.device)
    check(lib.ttsk_bn_bwd_apply(_ptr(dout), f32, _ptr(x), int(x.dtype == torch.float32), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), rows, Cn,
                                int(use_tanh), p, site, _ptr(rng), _ptr(sums), _ptr(dx), _ptr(dgamma), _ptr(dbeta), int(accumulate), *_lim(frame_limit),
                                _stream()), "ttsk_bn_bwd_apply")
    return dx


# ---------------------------------------------------------------------------------------------------- loss / optimiser

def fs2_loss(mel, post, mel_t, mel_lens, pitch, energy, logd, pitch_t, energy_t, dur_t, src_lens, grad_scale=1.0, frame_limit=None):
    """Returns (losses[8] fp32 device, dmel_sum, dpost, dpitch, denergy, dlogd).  See include/ttsk.h."""
    _dev(mel, post, mel_t, mel_lens, pitch, energy, logd, pitch_t, energy_t, dur_t, src_lens)
    B, T, nm = mel.shape
    Tt = mel_t.shape[1]
    Lp = pitch.shape[1]
    dev = mel.device
    lib = L.load()
    dmel, dpost = _f32(B, T, nm, device=dev), _f32(B, T, nm, device=dev)
    dstack = _f32(3, B, Lp, device=dev)          # (duration, pitch, energy): the order of the grouped predictor backward
    dd, dp, de = dstack[0], dstack[1], dstack[2]
    partials = _f32(lib.ttsk_fs2_loss_nblocks(), 6, device=dev)
    losses = _f32(8, device=dev)
    check(lib.ttsk_fs2_loss(_ptr(mel), _ptr(post), _ptr(mel_t), _ptr(mel_lens), _ptr(pitch), _ptr(energy), _ptr(logd),
                            _ptr(pitch_t), _ptr(energy_t), _ptr(dur_t), _ptr(src_lens), B, T, Tt, nm, Lp, grad_scale,
                            _ptr(dmel), _ptr(dpost), _ptr(dp), _ptr(de), _ptr(dd), _ptr(partials), _ptr(losses), _lim(frame_limit)[0], _stream()),
          "ttsk_fs2_loss")
    return losses, dmel, dpost, dp, de, dd


def fs2_loss_split(mel, post, mel_t, mel_lens, pitch, energy, logd, pitch_t, energy_t, dur_t, src_lens, side, grad_scale=1.0, frame_limit=None):
    """fs2_loss for a step whose variance predictors ran on stream `side` and nobody has waited for it: the frame-level terms and dmel_sum /
    dpost on the current stream (which does NOT wait for `side`), the phoneme-level terms and dpitch / denergy / dlogd on `side`, right
    behind the predictors (they need nothing of this stream).  The eight loss values come from fs2_loss_finalize, on any stream that has
    waited for both.  Same values, bit for bit, as fs2_loss (include/ttsk.h: ttsk_fs2_loss_mel / _var / _finalize)."""
    _dev(mel, post, mel_t, mel_lens, pitch, energy, logd, pitch_t, energy_t, dur_t, src_lens)
    B, T, nm = mel.shape
    Tt = mel_t.shape[1]
    Lp = pitch.shape[1]
    dev = mel.device
    lib = L.load()
    dmel, dpost = _f32(B, T, nm, device=dev), _f32(B, T, nm, device=dev)
    partials = _f32(lib.ttsk_fs2_loss_nblocks(), 6, device=dev)
    losses = _f32(8, device=dev)
    lim = _lim(frame_limit)[0]
    check(lib.ttsk_fs2_loss_mel(_ptr(mel), _ptr(post), _ptr(mel_t), _ptr(mel_lens), B, T, Tt, nm, grad_scale, _ptr(dmel), _ptr(dpost), _ptr(partials),
                                lim, _stream()), "ttsk_fs2_loss_mel")
    with torch.cuda.stream(side):
        # what `side` writes is allocated under `side`: the allocator hands a stream memory whose earlier users ran on that stream (a buffer of
        # the current stream's pool may still hold a live activation when `side` — far ahead of it — gets here; so it is in a captured step)
        dstack = _f32(3, B, Lp, device=dev)
        dd, dp, de = dstack[0], dstack[1], dstack[2]
        partials_var = _f32(lib.ttsk_fs2_loss_nblocks(), 6, device=dev)
        check(lib.ttsk_fs2_loss_var(_ptr(pitch), _ptr(energy), _ptr(logd), _ptr(pitch_t), _ptr(energy_t), _ptr(dur_t), _ptr(src_lens), B, Lp,
                                    grad_scale, _ptr(dp), _ptr(de), _ptr(dd), _ptr(partials_var), _stream()), "ttsk_fs2_loss_var")
    return losses, dmel, dpost, dp, de, dd, (partials, partials_var, lim, (B, T, nm))


def fs2_loss_finalize(losses, pending, src_lens):
    """The eight loss values of fs2_loss_split, on the current stream — which must have waited for both halves."""
    partials, partials_var, lim, (B, T, nm) = pending
    check(L.load().ttsk_fs2_loss_finalize(_ptr(partials), _ptr(partials_var), _ptr(src_lens), B, T, nm, lim, _ptr(losses), _stream()),
          "ttsk_fs2_loss_finalize")


def optim_state(device, seed=1234, sched_step=0):
    """Device state block (see include/ttsk.h): int64[8] view; fields set here, advanced by kernels."""
    n = L.load().ttsk_optim_state_bytes() // 8
    st = torch.zeros(n, dtype=torch.int64)
    st[0] = sched_step
    st[2] = seed
    return st.to(device)


def rng_of(state):
    return state[2:4]


def dropout_keep_mask(state, site, n, p):
    """uint8[n] keep-mask of dropout site `site` at the current (seed, step) of the device state block (ttsk_dropout_keep_mask): what
    every kernel of the step draws for that site; element order = row-major over the site's [rows][C] tensor."""
    keep = torch.empty(n, dtype=torch.uint8, device=state.device)
    check(L.load().ttsk_dropout_keep_mask(_ptr(rng_of(state)), int(site), int(n), float(p), _ptr(keep), _stream()), "ttsk_dropout_keep_mask")
    return keep


def optim_advance(state, d_model, warmup, anneal_steps, anneal_rate, beta1, beta2):
    arr = (C.c_float * 4)(*([float(a) for a in anneal_steps] + [0.0] * (4 - len(anneal_steps))))
    check(L.load().ttsk_optim_advance(_ptr(state), float(d_model), float(warmup), C.cast(arr, C.c_void_p), len(anneal_steps),
                                      anneal_rate, beta1, beta2, _stream()), "ttsk_optim_advance")


def rng_advance(state):
    check(L.load().ttsk_rng_advance(_ptr(state), _stream()), "ttsk_rng_advance")


def clip_adam_step(params, grads, m, v, shadow, state, partials, max_norm, beta1, beta2, eps, zero_grad=True):
    _dev(params, grads, m, v, shadow, state, partials)
    check(L.load().ttsk_clip_adam_step(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(shadow), params.numel(), _ptr(state),
                                       _ptr(partials), max_norm, beta1, beta2, eps, int(zero_grad), _stream()),
          "ttsk_clip_adam_step")


@_family("clip_adam", lambda *a, **kw: 0.0)
def optim_step(params, grads, m, v, shadow, state, partials, max_norm, beta1, beta2, eps, d_model, warmup, anneal_steps, anneal_rate,
               zero_grad=True, advance_rng=False):
    """optim_advance (+ rng_advance) + clip_adam_step in two launches (ttsk_optim_step)."""
    _dev(params, grads, m, v, shadow, state, partials)
    arr = (C.c_float * 4)(*([float(a) for a in anneal_steps] + [0.0] * (4 - len(anneal_steps))))
    check(L.load().ttsk_optim_step(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(shadow), params.numel(), _ptr(state), _ptr(partials),
                                   max_norm, beta1, beta2, eps, int(zero_grad), float(d_model), float(warmup), C.cast(arr, C.c_void_p),
                                   len(anneal_steps), anneal_rate, int(advance_rng), _stream()), "ttsk_optim_step")


def adam_pack_tables(items, n_flat, device):
    """Device tables for optim_step_packed.  items: [(flat offset, (Cs, K, Ds), plain pack tensor or None, transposed pack tensor or
    None)] — the window kernels' packed weights; everything else of [0, n_flat) becomes the gap ranges.  Returns a dict of the tensors /
    counts the call needs (the tensors must stay alive and the packs where they are), or None when a weight does not tile
    (Cs % 32, Ds % 256) or the ranges overlap."""
    items = sorted(items, key=lambda it: it[0])
    arr = (L.AdamItem * len(items))()
    tile0, pos, gaps = 0, 0, []
    for i, (off, (Cs, K, Ds), pk, pkt) in enumerate(items):
        if Cs % 32 or Ds % 256 or off % 4 or off < pos:
            return None
        if off > pos:
            gaps.append((pos, off))
        arr[i].off, arr[i].pack, arr[i].pack_t = off, _ptr(pk), _ptr(pkt)
        arr[i].Cs, arr[i].K, arr[i].Ds, arr[i].tile0 = Cs, K, Ds, tile0
        tile0 += K * (Cs // 32) * (Ds // 256)
        pos = off + Cs * K * Ds
    if pos < n_flat:
        gaps.append((pos, n_flat))
    g, acc = [], 0
    for a, b in gaps:
        if (b - a) % 4:
            return None
        g += [a, b, acc]
        acc += (b - a) // 4
    dev_items = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.int64).clone().to(device)
    dev_gaps = torch.tensor(g if g else [0, 0, 0], dtype=torch.int64).to(device)
    return {"items": dev_items, "n_items": len(items), "n_tiles": tile0, "gaps": dev_gaps, "n_gaps": len(gaps), "gap_floats": acc * 4}


@_family("clip_adam", lambda *a, **kw: 0.0)
def optim_step_packed(params, grads, m, v, shadow, state, partials, max_norm, beta1, beta2, eps, d_model, warmup, anneal_steps, anneal_rate,
                      tables, zero_grad=True, advance_rng=False):
    """optim_step whose Adam launch also writes the window kernels' weight packs (ttsk_optim_step_packed): no pack launch afterwards."""
    _dev(params, grads, m, v, shadow, state, partials, tables["items"], tables["gaps"])
    arr = (C.c_float * 4)(*([float(a) for a in anneal_steps] + [0.0] * (4 - len(anneal_steps))))
    check(L.load().ttsk_optim_step_packed(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(shadow), params.numel(), _ptr(state), _ptr(partials),
                                          max_norm, beta1, beta2, eps, int(zero_grad), float(d_model), float(warmup), C.cast(arr, C.c_void_p),
                                          len(anneal_steps), anneal_rate, int(advance_rng), _ptr(tables["items"]), tables["n_items"],
                                          tables["n_tiles"], _ptr(tables["gaps"]), tables["n_gaps"], tables["gap_floats"], _stream()),
          "ttsk_optim_step_packed")


# ---------------------------------------------------------------------------------------------------- HiFi-GAN helpers

def weight_norm_fold(v, g):
    """w = g * v / ||v|| with the norm over every dim but 0 (reference: hifi/models.py:203-210)."""
    _dev(v, g)
    v = v.contiguous()
    rows = v.shape[0]
    w = torch.empty_like(v)
    check(L.load().ttsk_weight_norm_fold(_ptr(v), _ptr(g.contiguous()), _ptr(w), rows, v.numel() // rows, _stream()),
          "ttsk_weight_norm_fold")
    return w


def pack_conv_weight(w, transposed=False, dtype=bf16):
    """Conv1d (Cout,Cin,k) fp32 -> (Cout,k,Cin) 16-bit, or ConvTranspose1d (Cin,Cout,k) fp32 -> (k,Cout,Cin) 16-bit."""
    _dev(w)
    d0, d1, d2 = w.shape
    shape = (d2, d1, d0) if transposed else (d0, d2, d1)
    out = torch.empty(shape, dtype=dtype, device=w.device)
    check(L.load().ttsk_pack_conv_weight(_ptr(w.contiguous()), _ptr(out), int(dtype == f16), d0, d1, d2, int(transposed), _stream()),
          "ttsk_pack_conv_weight")
    return out


def avg3(a, b, c, scale, out=None, slope=1.0):
    """out = leaky_relu((a + b + c) * scale, slope)  (slope 1.0 = no activation)."""
    _dev(a, b, c)
    if out is None:
        out = torch.empty_like(a)
    check(L.load().ttsk_avg3(_ptr(a), _ptr(b), _ptr(c), _ptr(out), int(a.dtype == f16), a.numel(), scale, slope, _stream()), "ttsk_avg3")
    return out


def hifi_resblock1_supported(C_, K):
    return bool(L.load().ttsk_hifi_resblock1_supported(C_, K))


def pack_resblock_weight(w, dtype=f16):
    """Folded Conv1d weight (C, C, K) fp32 -> MFMA-fragment-major 16-bit pack for ttsk_hifi_resblock1."""
    _dev(w)
    Cn, _, K = w.shape
    out = torch.empty(L.load().ttsk_resblock_pack_elems(Cn, K), dtype=dtype, device=w.device)
    check(L.load().ttsk_pack_resblock_weight(_ptr(w.contiguous()), _ptr(out), int(dtype == f16), Cn, K, _stream()),
          "ttsk_pack_resblock_weight")
    return out


def hifi_resblock1(x, weights, biases, dilations, out, K, mode=0, scale=1.0, slope=0.1, final_slope=1.0):
    """Fused ResBlock1 (reference: hifi/models.py:88-95).  x/out (B, len, C) 16-bit; weights: six fragment-major packs
    (pack_resblock_weight) and biases six fp32 (C,) in the order convs1[0], convs2[0], convs1[1], convs2[1], convs1[2],
    convs2[2].  mode 0: out = y, 1: out += y, 2: out = (out + y) * scale; final_slope: LeakyReLU on the stored value."""
    _dev(x, out, *weights, *biases)
    Bn, ln, Cn = x.shape
    wp = (C.c_void_p * 6)(*[w.data_ptr() for w in weights])
    bp = (C.c_void_p * 6)(*[b.data_ptr() for b in biases])
    dl = (C.c_int32 * 3)(*[int(d) for d in dilations])
    check(L.load().ttsk_hifi_resblock1(_ptr(x), _ptr(out), int(x.dtype == f16), C.cast(wp, C.c_void_p), C.cast(bp, C.c_void_p),
                                       C.cast(dl, C.c_void_p), Bn, ln, Cn, K, mode, scale, slope, final_slope, _stream()),
          "ttsk_hifi_resblock1")
    return out


def hifi_mrf32_post_supported(C_, ks, k_post):
    return len(ks) == 3 and bool(L.load().ttsk_hifi_mrf32_post_supported(C_, int(ks[0]), int(ks[1]), int(ks[2]), int(k_post)))


def hifi_mrf32_post(x, weights, biases, dilations, ks, w_post, b_post, slope=0.1, final_slope=0.01, scale=1.0 / 3.0, stage_out=None):
    """The generator's whole last stage in one launch (reference: hifi/models.py:190-199; csrc/mrf32.hip): x (B, len, 32) 16-bit raw
    stage input -> (B, 1, len) fp32 waveform = tanh(conv_post(leaky_relu(mean_j ResBlock1_j(x), 0.01))).  weights / biases: 18 packs /
    (32,) fp32 vectors, block j's six convs at [6 j ..] in hifi_resblock1's order; dilations: three triples; ks: the three kernel sizes;
    w_post (1, 7, 32) tap-major 16-bit.  stage_out (optional, like x): receives the activated average conv_post reads."""
    _dev(x, w_post, b_post, stage_out, *weights, *biases)
    Bn, ln, Cn = x.shape
    assert len(weights) == 18 and len(biases) == 18 and x.is_contiguous()
    wp = (C.c_void_p * 18)(*[w.data_ptr() for w in weights])
    bp = (C.c_void_p * 18)(*[b.data_ptr() for b in biases])
    dl = (C.c_int32 * 9)(*[int(d) for tri in dilations for d in tri])
    out = torch.empty(Bn, 1, ln, dtype=torch.float32, device=x.device)
    check(L.load().ttsk_hifi_mrf32_post(_ptr(x), _ptr(out), _ptr(stage_out), int(x.dtype == f16), C.cast(wp, C.c_void_p), C.cast(bp, C.c_void_p),
                                        C.cast(dl, C.c_void_p), _ptr(w_post), _ptr(b_post), Bn, ln, Cn, int(ks[0]), int(ks[1]), int(ks[2]),
                                        int(w_post.shape[-2]), slope, final_slope, scale, _stream()), "ttsk_hifi_mrf32_post")
    return out


def hifi_upsample2_supported(Cin, Cout, stride, k):
    return bool(L.load().ttsk_hifi_upsample2_supported(Cin, Cout, stride, k))


def hifi_upsample2(x, Wp, bias):
    """ConvTranspose1d(stride 2, kernel 4, padding 1): x (B, T, Cin) 16-bit, Wp (4, Cout, Cin) -> (B, 2T, Cout)."""
    _dev(x, Wp, bias)
    Bsz, T, Cin = x.shape
    Cout = Wp.shape[1]
    out = torch.empty(Bsz, 2 * T, Cout, dtype=x.dtype, device=x.device)
    check(L.load().ttsk_hifi_upsample2(_ptr(x), _ptr(Wp), _ptr(bias), _ptr(out), int(x.dtype == f16), Bsz, T, Cin, Cout, _stream()),
          "ttsk_hifi_upsample2")
    return out


def hifi_upsample_win_supported(Cin, Cout, stride, k):
    return k == 2 * stride and bool(L.load().ttsk_hifi_upsample_win_supported(Cin, Cout, stride))


def hifi_upsample8_supported(Cin, Cout, stride, k):
    return stride == 8 and hifi_upsample_win_supported(Cin, Cout, stride, k)


def hifi_upsample_win_pack(Wp, bias, stride):
    """Wp (2*stride, Cout, Cin) 16-bit (pack_conv_weight(..., transposed=True) of a ConvTranspose1d weight, padding stride/2) ->
    (fragment-major pack of the (stride*Cout, 2, Cin) pseudo-weight, bias repeated for the phases): output frame stride*t + r =
    x[t] . w[r + h] + (r < h ? x[t - 1] . w[r + h + stride] : x[t + 1] . w[r - h]), h = stride / 2 — include/ttsk.h:ttsk_hifi_upsample_win."""
    _dev(Wp, bias)
    k, Cout, Cin = Wp.shape
    h = stride // 2
    r = torch.arange(stride, device=Wp.device)
    slot0 = Wp[r + h]                                             # (stride, Cout, Cin)
    slot1 = Wp[torch.where(r < h, r + h + stride, r - h)]
    W2 = torch.stack([slot0, slot1], dim=2).reshape(stride * Cout, 2, Cin).contiguous()      # [(r, co)][slot][ci]
    pack = torch.empty(W2.numel(), dtype=Wp.dtype, device=Wp.device)
    win_conv_pack_items([(W2, pack, False)])
    return pack, bias.float().repeat(stride).contiguous()


def hifi_upsample8_pack(Wp, bias):
    return hifi_upsample_win_pack(Wp, bias, 8)


def hifi_upsample_win(x, pack, bias_rep, Cout, stride):
    """ConvTranspose1d(kernel 2*stride, padding stride/2) on the window-conv kernel: x (B, T, Cin) fp16 -> (B, stride*T, Cout)."""
    _dev(x, pack, bias_rep)
    Bsz, T, Cin = x.shape
    out = torch.empty(Bsz, stride * T, Cout, dtype=x.dtype, device=x.device)
    check(L.load().ttsk_hifi_upsample_win(_ptr(x), _ptr(pack), _ptr(bias_rep), _ptr(out), int(x.dtype == f16), Bsz, T, Cin, Cout, stride,
                                          _stream()), "ttsk_hifi_upsample_win")
    return out


def hifi_conv_pre_win_supported(Cin, Cout, k):
    return bool(L.load().ttsk_hifi_conv_pre_win_supported(Cin, Cout, k))


def hifi_conv_pre_win_pack(w16):
    """(Cout, k, 80) fp16 tap-major weight (pack_conv_weight) -> the window kernel's fragment-major pack (contraction padded to 96)."""
    _dev(w16)
    Cout, k, Cin = w16.shape
    pack = torch.empty(win_pack_numel(Cout, k, Cin, False), dtype=w16.dtype, device=w16.device)
    win_conv_pack_items([(w16.contiguous(), pack, False)])
    return pack


def hifi_conv_pre_win(x, pack, bias, Cout, k, slope):
    """lrelu(Conv1d(80 -> Cout, k)(x) + bias, slope) on the window-conv kernel: x (B, T, 80) fp16 -> (B, T, Cout) fp16."""
    _dev(x, pack, bias)
    Bsz, T, Cin = x.shape
    out = torch.empty(Bsz, T, Cout, dtype=x.dtype, device=x.device)
    check(L.load().ttsk_hifi_conv_pre_win(_ptr(x), _ptr(pack), _ptr(bias), _ptr(out), int(x.dtype == f16), Bsz, T, Cin, Cout, k, slope, _stream()),
          "ttsk_hifi_conv_pre_win")
    return out


def hifi_upsample_loop_supported(Cin, Cout, stride, k):
    return k == 2 * stride and bool(L.load().ttsk_hifi_upsample_loop_supported(Cin, Cout, stride))


def hifi_upsample_loop(x, pack, bias_rep, Cout, stride):
    """hifi_upsample_win's operator and operands on the kernel that loops over the channel groups per frame tile (Cin = 256, stride 8)."""
    _dev(x, pack, bias_rep)
    Bsz, T, Cin = x.shape
    out = torch.empty(Bsz, stride * T, Cout, dtype=x.dtype, device=x.device)
    check(L.load().ttsk_hifi_upsample_loop(_ptr(x), _ptr(pack), _ptr(bias_rep), _ptr(out), int(x.dtype == f16), Bsz, T, Cin, Cout, stride,
                                           _stream()), "ttsk_hifi_upsample_loop")
    return out


def hifi_upsample8(x, pack, bias8, Cout):
    return hifi_upsample_win(x, pack, bias8, Cout, 8)


def hifi_conv_post(x, w, bias):
    """tanh(Conv1d(C -> 1, k)(x)): x (B, len, C) 16-bit already activated, w (1, k, C) 16-bit -> (B, 1, len) fp32.
    reference: hifi/models.py:198-199."""
    _dev(x, w, bias)
    Bn, ln, Cn = x.shape
    out = torch.empty(Bn, 1, ln, dtype=torch.float32, device=x.device)
    check(L.load().ttsk_hifi_conv_post(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), int(x.dtype == f16), Bn, ln, Cn, w.shape[1], _stream()),
          "ttsk_hifi_conv_post")
    return out


def hifi_conv_window_supported(Cn, K, dil):
    return bool(L.load().ttsk_hifi_conv_window_supported(Cn, K, dil))


def hifi_conv_pair_supported(Cn, K, dil):
    return bool(L.load().ttsk_hifi_conv_pair_supported(Cn, K, dil))


def hifi_conv_pair_ws_tile(Cn):
    """Frames per tile of ttsk_hifi_conv_pair_ws at Cn channels (csrc/pairws.hip WsGeom::TT)."""
    return 192 if Cn <= 64 else 96


def hifi_conv_pair_ws_supported(Cn, K, dil, ln=0):
    """The weights-stationary pair kernel (csrc/pairws.hip) covers (C, K, dil) and utterances of `ln` frames (32-bit buffer offsets per utterance)."""
    return bool(L.load().ttsk_hifi_conv_pair_ws_supported(Cn, K, dil)) and ln * Cn * 2 < 2 ** 30


def hifi_conv_pair(x, w1_pack, bias1, w2_pack, bias2, K, dilation, slope=0.1, out=None, mode=0, scale=1.0, final_slope=1.0, ws=False, max_wgs=0):
    """y = c2(lrelu(c1_{K,dil}(lrelu(x)) + b1)) + b2 + x in one launch (C = 128; hifi/models.py:88-95, one dilation of ResBlock1).
    mode 0: out = y; 1: out += y; 2: out = lrelu((out + y) * scale, final_slope) (the MRF average, :190-197; ttsk_hifi_resblock1's modes).
    ws: the weights-stationary persistent kernel (C = 64, or C = 128 with K = 3; ttsk_hifi_conv_pair_ws, bit-identical)."""
    _dev(x, w1_pack, bias1, w2_pack, bias2, out)
    Bn, ln, Cn = x.shape
    if out is None:
        if mode >= 1:
            raise L.TtskError("hifi_conv_pair: mode %d accumulates into `out`" % mode)
        out = torch.empty_like(x)
    if ws:
        check(L.load().ttsk_hifi_conv_pair_ws(_ptr(x.contiguous()), _ptr(w1_pack), _ptr(bias1), _ptr(w2_pack), _ptr(bias2), _ptr(out), int(x.dtype == f16),
                                              Bn, ln, Cn, K, dilation, slope, mode, scale, final_slope, int(max_wgs), _stream()), "ttsk_hifi_conv_pair_ws")
        return out
    check(L.load().ttsk_hifi_conv_pair(_ptr(x), _ptr(w1_pack), _ptr(bias1), _ptr(w2_pack), _ptr(bias2), _ptr(out), int(x.dtype == f16), Bn,
                                       ln, Cn, K, dilation, slope, mode, scale, final_slope, _stream()), "ttsk_hifi_conv_pair")
    return out


def hifi_conv_window(x, w_pack, bias, K, dilation=1, R=None, out2=None, lrelu_out=False, slope=0.1):
    """out = [lrelu](conv_{K,dil}(x) + bias [+ R]) on the window kernel (C = 128); out2 (optional tensor) = lrelu(out)."""
    _dev(x, w_pack, bias, R, out2)
    Bn, ln, Cn = x.shape
    out = torch.empty_like(x)
    check(L.load().ttsk_hifi_conv_window(_ptr(x), _ptr(w_pack), _ptr(bias), _ptr(R), _ptr(out), _ptr(out2), int(x.dtype == f16), Bn, ln,
                                         Cn, K, dilation, int(lrelu_out), slope, _stream()), "ttsk_hifi_conv_window")
    return out


# ---------------------------------------------------------------------------------------------------- mel extraction (f-3)

def stft_frames(wav, pad, rows, hop, scale):
    """wav (B, len) fp32 -> fp16 (B, rows, 3*hop) = [hi | hi | lo]: reflect-padded, scaled hop-block rows split as hi + lo."""
    _dev(wav)
    Bsz, n = wav.shape
    out = torch.empty(Bsz, rows, 3 * hop, dtype=f16, device=wav.device)
    check(L.load().ttsk_stft_frames(_ptr(wav), _ptr(out), Bsz, n, pad, rows, hop, scale, _stream()), "ttsk_stft_frames")
    return out


def mel_from_spec(spec, Bsz, rows, T, nbins, vals, start, off, nnz, n_mels, eps, clip=1e-5):
    """spec (B*rows, ld) fp32 [Re | Im] -> log-mel (B, n_mels, T), energy (B, T)."""
    _dev(spec, vals, start, off)
    mel = torch.empty(Bsz, n_mels, T, dtype=torch.float32, device=spec.device)
    energy = torch.empty(Bsz, T, dtype=torch.float32, device=spec.device)
    check(L.load().ttsk_mel_from_spec(_ptr(spec), spec.stride(0), _ptr(vals), _ptr(start), _ptr(off), nnz, _ptr(mel),
                                      _ptr(energy), Bsz, rows, T, nbins, n_mels, eps, clip, _stream()), "ttsk_mel_from_spec")
    return mel, energy
