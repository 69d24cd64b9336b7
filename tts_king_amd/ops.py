"""Thin torch-tensor -> C-ABI adapters (plumbing only: device pointers, strides, the current HIP stream).

Every function here enqueues hand-written gfx950 kernels from libttsk_hip.so on torch's current stream and
returns immediately; nothing is computed by PyTorch.  Tensors must live on a HIP device ("cuda:N" in
PyTorch-ROCm naming) — CPU tensors are rejected, there is no fallback.
"""
import ctypes as C

import torch

from . import lib as L
from .lib import (A_TR, B_TR, C_F32, RELU, ADD_R, R_F32, MASK_G, LRELU_IN, TANH, ACCUM_C, LRELU_OUT,  # noqa: F401
                  GemmDesc, check)

bf16 = torch.bfloat16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.TtskError("tts_king_amd kernels need HIP device tensors (got %s); there is no CPU path" % t.device)


def _ptr(t):
    return None if t is None else t.data_ptr()


def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, flags=0, alpha=1.0, bias=None, R=None, ldr=0, G=None, ldg=0, C2=None,
         nz1=1, nz2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), taps=0, seg_len=0, tap_shift0=0, tap_dshift=0,
         b_tap_stride=0, bseg_len=0, bshift0=0, bdshift=0, out_seg=0, out_mul=0, out_add=0, splits=1, sCs=0,
         in_slope=0.0, out_slope=0.0):
    """Raw descriptor-level call of ttsk_gemm (see include/ttsk.h).  A/B/Cout may be views: the data pointer of
    the view is the operand origin."""
    _dev(A, B, Cout, bias, R, G, C2)
    d = GemmDesc()
    d.A, d.B, d.C, d.C2 = _ptr(A), _ptr(B), _ptr(Cout), _ptr(C2)
    d.bias, d.R, d.G = _ptr(bias), _ptr(R), _ptr(G)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldr, d.ldg = lda, ldb, ldc, ldr, ldg
    if Cout.dtype == torch.float32:
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
    d.out_seg, d.out_mul, d.out_add = out_seg, out_mul, out_add
    d.splits, d.sCs = splits, sCs
    check(L.load().ttsk_gemm(C.byref(d), _stream()), "ttsk_gemm")
    return Cout


def reduce_slabs(slabs, n_slabs, slab_stride, dst, numel, accumulate=False):
    _dev(slabs, dst)
    check(L.load().ttsk_reduce_slabs(_ptr(slabs), n_slabs, slab_stride, _ptr(dst), numel, int(accumulate), _stream()),
          "ttsk_reduce_slabs")
    return dst


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

def _splits_for(M, N, K, nz=1):
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * nz
    kchunks = (K + 63) // 64
    return max(1, min(512 // max(tiles, 1), kchunks // 4))


def linear(x, W, bias=None, out=None, flags=0, out_dtype=bf16, R=None, G=None, C2=None, alpha=1.0, out_slope=0.0):
    """y[M,N] = x[M,K] @ W[N,K]^T (+bias, epilogue).  reference: nn.Linear sites SubLayers.py:41-43,62; fastspeech2.py:102."""
    M, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=x.device)
    return gemm(x, W, out, M, N, K, x.stride(0), W.stride(0), out.stride(0), flags=flags, alpha=alpha, bias=bias,
                R=R, ldr=0 if R is None else R.stride(0), G=G, ldg=0 if G is None else G.stride(0), C2=C2,
                out_slope=out_slope)


def linear_dx(dy, W, out=None, R=None, G=None, out_dtype=bf16):
    """dx[M,K] = dy[M,N] @ W[N,K]   (W read in place through the transposing LDS read)."""
    M, N = dy.shape
    K = W.shape[1]
    if out is None:
        out = torch.empty(M, K, dtype=out_dtype, device=dy.device)
    return gemm(dy, W, out, M, K, N, dy.stride(0), W.stride(0), out.stride(0), flags=B_TR, R=R,
                ldr=0 if R is None else R.stride(0), G=G, ldg=0 if G is None else G.stride(0))


def linear_dw(dy, x, dst, accumulate=True):
    """dst[N,K] (fp32) (+)= dy[M,N]^T @ x[M,K]: contraction over rows, split-K slabs + deterministic reduce."""
    M, N = dy.shape
    K = x.shape[1]
    splits = _splits_for(N, K, M)
    if splits == 1 and not accumulate:
        return gemm(dy, x, dst, N, K, M, dy.stride(0), x.stride(0), K, flags=A_TR | B_TR)
    slabs = torch.empty(splits, N * K, dtype=torch.float32, device=dy.device)
    gemm(dy, x, slabs, N, K, M, dy.stride(0), x.stride(0), K, flags=A_TR | B_TR, splits=splits, sCs=N * K)
    return reduce_slabs(slabs, splits, N * K, dst, N * K, accumulate)


def conv1d(x, W, bias, dilation=1, out=None, flags=0, out_dtype=bf16, R=None, C2=None, in_slope=0.0, out_slope=0.0,
           alpha=1.0):
    """'same' Conv1d on channels-last activations.  x (B,T,Cin) bf16, W (Cout,k,Cin) bf16 -> (B,T,Cout).
    reference: SubLayers.py:96 (k=9/1), modules.py:337-355 (k=3), Layers.py:59-67 (k=5), hifi/models.py:88-95."""
    Bsz, T, Cin = x.shape
    Cout, k, _ = W.shape
    pad = dilation * (k - 1) // 2
    if out is None:
        out = torch.empty(Bsz, T, Cout, dtype=out_dtype, device=x.device)
    gemm(x, W, out, Bsz * T, Cout, Cin, Cin, k * Cin, Cout, flags=flags, bias=bias, R=R, ldr=Cout, C2=C2, taps=k,
         seg_len=T, tap_shift0=-pad, tap_dshift=dilation, b_tap_stride=Cin, in_slope=in_slope, out_slope=out_slope,
         alpha=alpha)
    return out


def conv1d_dx(dy, W, dilation=1, out=None, R=None, G=None):
    """dx (B,T,Cin) = sum_j dy[t + pad - j*dil] @ W[:, j, :]."""
    Bsz, T, Cout = dy.shape
    _, k, Cin = W.shape
    pad = dilation * (k - 1) // 2
    if out is None:
        out = torch.empty(Bsz, T, Cin, dtype=bf16, device=dy.device)
    gemm(dy, W, out, Bsz * T, Cin, Cout, Cout, k * Cin, Cin, flags=B_TR, R=R, ldr=Cin, G=G, ldg=Cin, taps=k, seg_len=T,
         tap_shift0=pad, tap_dshift=-dilation, b_tap_stride=Cin)
    return out


def conv1d_dw(dy, x, dst, dilation=1, k=1, accumulate=True):
    """dst (Cout,k,Cin) fp32 (+)= sum_rows dy[r, co] * x[r + j*dil - pad, ci]."""
    Bsz, T, Cout = dy.shape
    Cin = x.shape[2]
    pad = dilation * (k - 1) // 2
    rows = Bsz * T
    splits = _splits_for(Cout, Cin, rows, k)
    n = Cout * k * Cin
    if splits == 1 and not accumulate:
        slabs = dst
    else:
        slabs = torch.empty(splits, n, dtype=torch.float32, device=dy.device)
    gemm(dy, x, slabs, Cout, Cin, rows, Cout, Cin, k * Cin, flags=A_TR | B_TR, nz2=k, sC=(0, Cin), bseg_len=T,
         bshift0=-pad, bdshift=dilation, splits=splits, sCs=n)
    if slabs is not dst:
        reduce_slabs(slabs, splits, n, dst, n, accumulate)
    return dst


def conv_transpose1d(x, Wp, bias, stride, k, out=None, in_slope=0.0, flags=0):
    """ConvTranspose1d(padding=(k-stride)//2) as `stride` polyphase implicit GEMMs.
    x (B,T,Cin) bf16, Wp (k, Cout, Cin) bf16 (tap-major repack of torch's (Cin,Cout,k)) -> (B,T*stride,Cout).
    reference: hifi/models.py:166-176,189."""
    Bsz, T, Cin = x.shape
    Cout = Wp.shape[1]
    p = (k - stride) // 2
    if out is None:
        out = torch.empty(Bsz, T * stride, Cout, dtype=bf16, device=x.device)
    taps = k // stride
    for r in range(stride):
        qoff = max(0, -((r - p) // stride))          # ceil((p - r)/stride) clipped at 0
        gemm(x, Wp[r], out, Bsz * T, Cout, Cin, Cin, Cin, Cout, flags=flags | (LRELU_IN if in_slope else 0),
             bias=bias, taps=taps, seg_len=T, tap_shift0=qoff, tap_dshift=-1, b_tap_stride=stride * Cout * Cin,
             out_seg=T * stride, out_mul=stride, out_add=qoff * stride + r - p, in_slope=in_slope)
    return out
