"""GPU: the MFMA GEMM / implicit-conv kernel (tts_king_amd/csrc/gemm.hip) through the C ABI, against fp64
CPU math on the same bf16-rounded operands.  Integer-valued cases are exact (they pin the operand/fragment
layouts, including the ds_read_b64_tr_b16 transposed reads); random cases use a tolerance scaled by K
(bf16 products are exact in fp32, only the summation order differs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bf(t):
    return t.to(torch.bfloat16)


def rnd(*shape, seed=0, ints=False):
    g = torch.Generator().manual_seed(seed)
    if ints:
        return torch.randint(-3, 4, shape, generator=g).float()
    return torch.randn(*shape, generator=g)


def check(out, ref, K, exact=False, out_bf16=True):
    out = out.float().cpu().double()
    ref = ref.double()
    if exact:
        assert torch.equal(out, ref), float((out - ref).abs().max())
        return
    tol = 2e-6 * K ** 0.5 * float(ref.abs().max() + 1) + (float(ref.abs().max()) * 2 ** -8 if out_bf16 else 0)
    err = float((out - ref).abs().max())
    assert err <= tol, (err, tol)


@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 264), (1024, 768, 256), (77, 80, 256)])
def test_nt_bias_relu(M, N, K, ints):
    from tts_king_amd import ops
    a, w, b = bf(rnd(M, K, seed=1, ints=ints)), bf(rnd(N, K, seed=2, ints=ints)), rnd(N, seed=3, ints=ints)
    ref = torch.relu(a.double() @ w.double().t() + b.double())
    out32 = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), flags=ops.RELU, out_dtype=torch.float32)
    check(out32, ref, K, exact=ints, out_bf16=False)
    out16 = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), flags=ops.RELU)
    check(out16, bf(ref.float()).double() if ints else ref, K, exact=ints)


@pytest.mark.parametrize("ints", [True, False])
def test_batched_scores_like(ints):
    """S[z] = alpha * Q[z] K[z]^T with two-level batch strides into a fused [rows][768] QKV buffer; N = 423."""
    from tts_king_amd import ops
    Bsz, H, S, dk = 3, 2, 423, 128
    qkv = bf(rnd(Bsz * S, 3 * H * dk, seed=4, ints=ints)).to(DEV)
    Sp = (S + 7) // 8 * 8
    out = torch.zeros(Bsz * H, S, Sp, dtype=torch.float32, device=DEV)
    alpha = 0.5 if ints else dk ** -0.5
    ops.gemm(qkv, qkv[:, H * dk:], out, S, S, dk, 3 * H * dk, 3 * H * dk, Sp, alpha=alpha, nz1=Bsz, nz2=H,
             sA=(S * 3 * H * dk, dk), sB=(S * 3 * H * dk, dk), sC=(H * S * Sp, S * Sp))
    q = qkv.cpu().double().view(Bsz, S, 3, H, dk)
    ref = alpha * torch.einsum("bqhd,bkhd->bhqk", q[:, :, 0], q[:, :, 1]).reshape(Bsz * H, S, S)
    check(out[:, :, :S], ref, dk, exact=ints, out_bf16=False)
    assert float(out[:, :, S:].abs().max()) == 0.0


@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("M,N,K", [(423, 128, 423), (200, 256, 768), (128, 80, 64)])
def test_b_transposed(M, N, K, ints):
    """C = A[M,K] @ B[K,N] with B read in place (P·V, dX = dY·W)."""
    from tts_king_amd import ops
    Kp = (K + 7) // 8 * 8
    a = torch.zeros(M, Kp)
    a[:, :K] = rnd(M, K, seed=5, ints=ints)
    a, b = bf(a), bf(rnd(K, N, seed=6, ints=ints))
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(a.to(DEV), b.to(DEV), out, M, N, K, Kp, N, N, flags=ops.B_TR)
    check(out, a[:, :K].double() @ b.double(), K, exact=ints, out_bf16=False)


@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("M,N,K", [(256, 256, 1000), (768, 256, 6768), (423, 128, 423), (80, 512, 300)])
def test_both_transposed_splitk(M, N, K, ints):
    """C = A[K,M]^T @ B[K,N] (dW, dK, dV) incl. split-K slabs + reduce with accumulate."""
    from tts_king_amd import ops
    Mp = (M + 7) // 8 * 8
    a = torch.zeros(K, Mp)
    a[:, :M] = rnd(K, M, seed=7, ints=ints)
    a, b = bf(a), bf(rnd(K, N, seed=8, ints=ints))
    ref = a[:, :M].double().t() @ b.double()
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(a.to(DEV), b.to(DEV), out, M, N, K, Mp, N, N, flags=ops.A_TR | ops.B_TR)
    check(out, ref, K, exact=ints, out_bf16=False)
    if Mp == M:
        dst = torch.ones(M, N, dtype=torch.float32, device=DEV)
        ops.linear_dw(a.to(DEV), b.to(DEV), dst, accumulate=True)
        check(dst, ref + 1, K, exact=ints, out_bf16=False)


@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("Bsz,T,Cin,Cout,k,dil", [(3, 50, 256, 1024, 9, 1), (2, 423, 80, 512, 5, 1), (2, 131, 128, 128, 7, 3),
                                                  (1, 300, 32, 32, 11, 5), (2, 64, 256, 256, 3, 1), (2, 40, 1024, 256, 1, 1)])
def test_conv1d_fwd_dx_dw(Bsz, T, Cin, Cout, k, dil, ints):
    from tts_king_amd import ops
    x = bf(rnd(Bsz, T, Cin, seed=9, ints=ints))
    w = bf(rnd(Cout, Cin, k, seed=10, ints=ints) * (1.0 if ints else (Cin * k) ** -0.5))
    b = rnd(Cout, seed=11, ints=ints)
    dy = bf(rnd(Bsz, T, Cout, seed=12, ints=ints))
    pad = dil * (k - 1) // 2
    xd = x.double().transpose(1, 2).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y = F.conv1d(xd, wd, b.double(), dilation=dil, padding=pad)
    y.backward(dy.double().transpose(1, 2))
    wk = w.permute(0, 2, 1).contiguous().to(DEV)           # (Cout, k, Cin)
    out = ops.conv1d(x.to(DEV), wk, b.to(DEV), dilation=dil, out_dtype=torch.float32)
    check(out, y.detach().transpose(1, 2), Cin * k, exact=ints, out_bf16=False)
    dx = torch.empty(Bsz, T, Cin, dtype=torch.float32, device=DEV)
    ops.conv1d_dx(dy.to(DEV), wk, dilation=dil, out=dx)
    check(dx, xd.grad.transpose(1, 2), Cout * k, exact=ints, out_bf16=False)
    dw = torch.zeros(Cout, k, Cin, dtype=torch.float32, device=DEV)
    ops.conv1d_dw(dy.to(DEV), x.to(DEV), dw, dilation=dil, k=k, accumulate=True)
    check(dw, wd.grad.permute(0, 2, 1), Bsz * T, exact=ints, out_bf16=False)


def test_conv_lrelu_in_and_epilogues():
    from tts_king_amd import ops
    Bsz, T, C, k, dil = 2, 100, 64, 3, 3
    x = bf(rnd(Bsz, T, C, seed=13))
    w = bf(rnd(C, C, k, seed=14) * (C * k) ** -0.5)
    b = rnd(C, seed=15)
    r = bf(rnd(Bsz, T, C, seed=16))
    xin = bf(F.leaky_relu(x.float(), 0.1))                 # kernel rounds lrelu(x) to bf16 before the MFMA
    ref = F.conv1d(xin.double().transpose(1, 2), w.double(), b.double(), dilation=dil, padding=dil).transpose(1, 2) + r.double()
    wk = w.permute(0, 2, 1).contiguous().to(DEV)
    c2 = torch.empty(Bsz, T, C, dtype=torch.bfloat16, device=DEV)
    out = ops.conv1d(x.to(DEV), wk, b.to(DEV), dilation=dil, flags=ops.LRELU_IN, in_slope=0.1, R=r.to(DEV),
                     out_dtype=torch.float32, C2=c2)
    check(out, ref, C * k, out_bf16=False)
    check(c2, ref, C * k)
    g = bf(rnd(Bsz * T, C, seed=17))
    out2 = ops.linear(x.view(-1, C).to(DEV), w[:, :, 0].contiguous().to(DEV), None, G=g.to(DEV), flags=ops.TANH,
                      out_dtype=torch.float32)
    ref2 = torch.tanh(torch.where(g.double() > 0, x.view(-1, C).double() @ w[:, :, 0].double().t(), torch.zeros(())))
    check(out2, ref2, C, out_bf16=False)


@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("Cin,Cout,k,s,T", [(512, 256, 16, 8, 37), (128, 64, 4, 2, 301), (64, 32, 4, 2, 128)])
def test_conv_transpose_polyphase(Cin, Cout, k, s, T, ints):
    from tts_king_amd import ops
    Bsz = 2
    x = bf(rnd(Bsz, T, Cin, seed=18, ints=ints))
    w = bf(rnd(Cin, Cout, k, seed=19, ints=ints) * (1.0 if ints else (Cin * k / s) ** -0.5))
    b = rnd(Cout, seed=20, ints=ints)
    xin = bf(F.leaky_relu(x.float(), 0.1)) if not ints else x
    ref = F.conv_transpose1d(xin.double().transpose(1, 2), w.double(), b.double(), stride=s, padding=(k - s) // 2).transpose(1, 2)
    wp = w.permute(2, 1, 0).contiguous().to(DEV)           # (k, Cout, Cin)
    out = ops.conv_transpose1d(x.to(DEV), wp, b.to(DEV), s, k, in_slope=0.0 if ints else 0.1)
    assert out.shape == (Bsz, T * s, Cout)
    check(out, bf(ref.float()).double() if ints else ref, Cin * k // s, exact=ints)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("splits", [1, 3, 16])
def test_split_k_with_fused_epilogue(splits, dtype):
    """Split-K through the workspace: the reducer applies bias + residual + ReLU gate + second output exactly like the
    single-pass epilogue (conv taps included), for bf16 and fp16 operands."""
    from tts_king_amd import ops
    Bsz, T, Cin, Cout, k = 2, 96, 1024, 256, 9
    x = rnd(Bsz, T, Cin, seed=30).to(dtype)
    w = (rnd(Cout, Cin, k, seed=31) * (Cin * k) ** -0.5).to(dtype)
    b = rnd(Cout, seed=32)
    r = rnd(Bsz, T, Cout, seed=33).to(dtype)
    ref = torch.relu(F.conv1d(x.double().transpose(1, 2), w.double(), b.double(), padding=4).transpose(1, 2) + r.double())
    wk = w.permute(0, 2, 1).contiguous().to(DEV)
    out = torch.empty(Bsz, T, Cout, dtype=torch.float32, device=DEV)
    c2 = torch.empty(Bsz, T, Cout, dtype=dtype, device=DEV)
    ops.gemm(x.to(DEV), wk, out, Bsz * T, Cout, Cin, Cin, k * Cin, Cout, flags=ops.RELU, bias=b.to(DEV), R=r.to(DEV), ldr=Cout,
             C2=c2, taps=k, seg_len=T, tap_shift0=-4, tap_dshift=1, b_tap_stride=Cin, splits=splits)
    check(out, ref, Cin * k, out_bf16=False)
    assert float((c2.float().cpu() - out.cpu()).abs().max()) <= float(ref.abs().max()) * 2 ** -8


def test_plan_fills_the_chip():
    """ttsk_gemm_plan: small problems are split along K, LRELU_IN forces the register-staged kernel."""
    from tts_king_amd import ops
    from tts_king_amd.lib import GemmDesc
    def mk(M, N, K, taps=0, flags=0, nz2=1):
        d = GemmDesc()
        d.A = d.B = d.C = 256
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.taps, d.flags, d.nz1, d.nz2 = M, N, K, K, K, N, taps, flags, 1, nz2
        d.seg_len = M
        return d
    k, sp, ws = ops.plan(mk(1024, 256, 1024, 9))
    assert sp >= 4 and ws == sp * 1024 * 256 * 4
    k, sp, ws = ops.plan(mk(196608, 128, 128, 11, flags=ops.LRELU_IN))
    assert k == 1 and sp == 1 and ws == 0
    k, sp, ws = ops.plan(mk(196608, 128, 128, 11))
    assert sp == 1


# ---- the 256x128 LDS-DMA kernel (kernel=2) on every operand layout, with tails in M, N and K
DMA_KERNELS = pytest.mark.parametrize("kern", [2])
@DMA_KERNELS
@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 200, 264), (1024, 768, 256), (77, 80, 256), (513, 129, 1000)])
def test_k2_nt_bias_relu(M, N, K, ints, kern):
    from tts_king_amd import ops
    a, w, b = bf(rnd(M, K, seed=1, ints=ints)), bf(rnd(N, K, seed=2, ints=ints)), rnd(N, seed=3, ints=ints)
    ref = torch.relu(a.double() @ w.double().t() + b.double())
    out32 = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), flags=ops.RELU, out_dtype=torch.float32, kernel=kern, splits=1)
    check(out32, ref, K, exact=ints, out_bf16=False)
    out16 = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), flags=ops.RELU, kernel=kern)
    check(out16, bf(ref.float()).double() if ints else ref, K, exact=ints)


@DMA_KERNELS
@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("M,N,K", [(423, 128, 423), (200, 256, 768), (600, 80, 64)])
def test_k2_b_transposed(M, N, K, ints, kern):
    from tts_king_amd import ops
    Kp = (K + 7) // 8 * 8
    a = torch.zeros(M, Kp)
    a[:, :K] = rnd(M, K, seed=5, ints=ints)
    a, b = bf(a), bf(rnd(K, N, seed=6, ints=ints))
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(a.to(DEV), b.to(DEV), out, M, N, K, Kp, N, N, flags=ops.B_TR, kernel=kern, splits=1)
    check(out, a[:, :K].double() @ b.double(), K, exact=ints, out_bf16=False)


@DMA_KERNELS
@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("M,N,K,splits", [(256, 256, 1000, 1), (768, 256, 6768, 5), (423, 128, 423, 1), (80, 512, 300, 2), (1024, 256, 2000, 0)])
def test_k2_both_transposed(M, N, K, splits, ints, kern):
    from tts_king_amd import ops
    Mp = (M + 7) // 8 * 8
    a = torch.zeros(K, Mp)
    a[:, :M] = rnd(K, M, seed=7, ints=ints)
    a, b = bf(a), bf(rnd(K, N, seed=8, ints=ints))
    ref = a[:, :M].double().t() @ b.double()
    out = torch.ones(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(a.to(DEV), b.to(DEV), out, M, N, K, Mp, N, N, flags=ops.A_TR | ops.B_TR | ops.ACCUM_C, kernel=kern, splits=splits)
    check(out, ref + 1, K, exact=ints, out_bf16=False)


@DMA_KERNELS
@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("Bsz,T,Cin,Cout,k,dil", [(3, 50, 256, 1024, 9, 1), (2, 423, 80, 512, 5, 1), (2, 131, 128, 128, 7, 3),
                                                  (2, 300, 256, 256, 11, 5), (2, 40, 1024, 256, 1, 1)])
def test_k2_conv1d_fwd_dx_dw(Bsz, T, Cin, Cout, k, dil, ints, kern):
    from tts_king_amd import ops
    x = bf(rnd(Bsz, T, Cin, seed=9, ints=ints))
    w = bf(rnd(Cout, Cin, k, seed=10, ints=ints) * (1.0 if ints else (Cin * k) ** -0.5))
    b = rnd(Cout, seed=11, ints=ints)
    dy = bf(rnd(Bsz, T, Cout, seed=12, ints=ints))
    pad = dil * (k - 1) // 2
    xd = x.double().transpose(1, 2).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y = F.conv1d(xd, wd, b.double(), dilation=dil, padding=pad)
    y.backward(dy.double().transpose(1, 2))
    wk = w.permute(0, 2, 1).contiguous().to(DEV)
    out = ops.conv1d(x.to(DEV), wk, b.to(DEV), dilation=dil, out_dtype=torch.float32, kernel=kern)
    check(out, y.detach().transpose(1, 2), Cin * k, exact=ints, out_bf16=False)
    dx = torch.empty(Bsz, T, Cin, dtype=torch.float32, device=DEV)
    ops.conv1d_dx(dy.to(DEV), wk, dilation=dil, out=dx, kernel=kern)
    check(dx, xd.grad.transpose(1, 2), Cout * k, exact=ints, out_bf16=False)
    dw = torch.zeros(Cout, k, Cin, dtype=torch.float32, device=DEV)
    ops.conv1d_dw(dy.to(DEV), x.to(DEV), dw, dilation=dil, k=k, accumulate=True, kernel=kern)
    check(dw, wd.grad.permute(0, 2, 1), Bsz * T, exact=ints, out_bf16=False)


@DMA_KERNELS
def test_k2_fp16_and_polyphase(kern):
    from tts_king_amd import ops
    Cin, Cout, k, s, T, Bsz = 256, 128, 16, 8, 70, 2
    x = rnd(Bsz, T, Cin, seed=18).half()
    w = (rnd(Cin, Cout, k, seed=19) * (Cin * k / s) ** -0.5).half()
    b = rnd(Cout, seed=20)
    ref = F.conv_transpose1d(x.double().transpose(1, 2), w.double(), b.double(), stride=s, padding=(k - s) // 2).transpose(1, 2)
    wp = w.permute(2, 1, 0).contiguous().to(DEV)
    out = ops.conv_transpose1d(x.to(DEV), wp, b.to(DEV), s, k, kernel=kern)
    assert out.dtype == torch.float16
    err = float((out.float().cpu().double() - ref).abs().max())
    assert err <= 2e-6 * (Cin * k // s) ** 0.5 * float(ref.abs().max() + 1) + float(ref.abs().max()) * 2 ** -10, err


@pytest.mark.parametrize("kern", [1, 2])
def test_grouped_launch_equals_individual_launches(kern):
    """ttsk_gemm_group_* (both tile configurations): weight-gradient GEMMs of different shapes (split-K with deferred reduce, accumulate, conv taps as
    batch) queued in a DeferQueue produce bit-identical results to the same calls launched one by one on kernel 1 with the
    same split factors; with the group's own split policy (fewer, longer K ranges) they agree with fp64."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(11)
    cases = [(1024, 256, 256, 1, 3), (6768, 768, 256, 1, 8), (423, 128, 64, 1, 1), (2048, 512, 80, 5, 4), (640, 256, 1024, 3, 2)]
    tensors = []
    for rows, cout, cin, k, sp in cases:
        Bsz = 2 if k > 1 else 1
        T = rows // Bsz
        dy = bf(torch.randn(Bsz, T, cout, generator=g)).to(DEV)
        x = bf(torch.randn(Bsz, T, cin, generator=g)).to(DEV)
        tensors.append((dy, x, cout, cin, k, sp))
    outs = {}
    for mode in ("single", "grouped", "grouped-auto"):
        q = ops.DeferQueue(group_gemms=(mode != "single"))
        res = []
        for dy, x, cout, cin, k, sp in tensors:
            dst = torch.full((cout, k, cin), 0.5, dtype=torch.float32, device=DEV)
            kw = dict(defer=q, kernel=kern) if mode != "grouped-auto" else dict(defer=q)
            if mode != "grouped-auto":
                kw["splits"] = sp
            if k == 1:
                ops.linear_dw(dy.view(-1, cout), x.view(-1, cin), dst.view(cout, cin), **kw)
            else:
                ops.conv1d_dw(dy, x, dst, k=k, **kw)
            res.append(dst)
        if mode == "grouped":
            assert len(q.group) == len(cases)
        ops.flush_deferred(q)
        torch.cuda.synchronize()
        outs[mode] = [r.cpu() for r in res]
    # the same group with its grid capped at 5 workgroups (ttsk_gemm_group_launch_capped: each workgroup walks tiles wg, wg + 5, ...;
    # honoured by the 256x128 configuration) on a second stream, then the reducers on the first: bit-identical again
    q = ops.DeferQueue(group_gemms=True)
    res = []
    for dy, x, cout, cin, k, sp in tensors:
        dst = torch.full((cout, k, cin), 0.5, dtype=torch.float32, device=DEV)
        kw = dict(defer=q, kernel=kern, splits=sp)
        if k == 1:
            ops.linear_dw(dy.view(-1, cout), x.view(-1, cin), dst.view(cout, cin), **kw)
        else:
            ops.conv1d_dw(dy, x, dst, k=k, **kw)
        res.append(dst)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.flush_deferred_gemms(q, max_wgs=5, small_too=True)          # small_too: the 128x128 problems go out (uncapped) as well
    assert not q.group and len(q) > 0                      # the GEMMs went out, their split-K slabs still wait for the reducer
    torch.cuda.current_stream().wait_stream(side)
    ops.flush_deferred(q)
    torch.cuda.synchronize()
    for a, r in zip(outs["single"], res):
        assert torch.equal(a, r.cpu())
    for a, b, c, (dy, x, cout, cin, k, sp) in zip(outs["single"], outs["grouped"], outs["grouped-auto"], tensors):
        assert torch.equal(a, b)
        if k == 1:
            ref = dy.view(-1, cout).double().cpu().t() @ x.view(-1, cin).double().cpu() + 0.5
            check(b.view(cout, cin).to(DEV), ref, dy.shape[0] * dy.shape[1], exact=False, out_bf16=False)
            check(c.view(cout, cin).to(DEV), ref, dy.shape[0] * dy.shape[1], exact=False, out_bf16=False)
        else:
            assert float((c - a).abs().max()) <= 1e-3 * float(a.abs().max())


# ---- the 64-row tile (kernel = 3): A untransposed, every epilogue, conv taps, B transposed, split-K, tails
@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("M,N,K", [(64, 128, 64), (300, 200, 264), (6768, 256, 256), (1024, 768, 256), (77, 80, 256), (513, 129, 1000)])
def test_k3_nt_bias_relu(M, N, K, ints):
    from tts_king_amd import ops
    a, w, b = bf(rnd(M, K, seed=1, ints=ints)), bf(rnd(N, K, seed=2, ints=ints)), rnd(N, seed=3, ints=ints)
    ref = torch.relu(a.double() @ w.double().t() + b.double())
    out32 = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), flags=ops.RELU, out_dtype=torch.float32, kernel=3, splits=1)
    check(out32, ref, K, exact=ints, out_bf16=False)
    out16 = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV), flags=ops.RELU, kernel=3)
    check(out16, bf(ref.float()).double() if ints else ref, K, exact=ints)


@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("M,N,K,splits", [(423, 128, 423, 1), (6768, 256, 768, 1), (600, 80, 64, 1), (1024, 256, 1024, 4)])
def test_k3_b_transposed_residual_gate(M, N, K, splits, ints):
    """dX-shaped problems: B read through the transposing LDS read, fp32 / bf16 residual, ReLU gate, optional split-K."""
    from tts_king_amd import ops
    Kp = (K + 7) // 8 * 8
    a = torch.zeros(M, Kp)
    a[:, :K] = rnd(M, K, seed=5, ints=ints)
    a, b = bf(a), bf(rnd(K, N, seed=6, ints=ints))
    r, g = bf(rnd(M, N, seed=7, ints=ints)), bf(rnd(M, N, seed=8))
    ref = (a[:, :K].double() @ b.double() + r.double()) * (g.double() > 0)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(a.to(DEV), b.to(DEV), out, M, N, K, Kp, N, N, flags=ops.B_TR, kernel=3, splits=splits, R=r.to(DEV), ldr=N, G=g.to(DEV), ldg=N)
    check(out, ref, K, exact=ints, out_bf16=False)


@pytest.mark.parametrize("ints", [True, False])
@pytest.mark.parametrize("Bsz,T,Cin,Cout,k,dil", [(3, 50, 256, 1024, 9, 1), (16, 64, 256, 256, 3, 1), (2, 423, 80, 512, 5, 1), (2, 131, 128, 128, 7, 3)])
def test_k3_conv1d_fwd_dx(Bsz, T, Cin, Cout, k, dil, ints):
    from tts_king_amd import ops
    x = bf(rnd(Bsz, T, Cin, seed=9, ints=ints))
    w = bf(rnd(Cout, Cin, k, seed=10, ints=ints) * (1.0 if ints else (Cin * k) ** -0.5))
    b = rnd(Cout, seed=11, ints=ints)
    dy = bf(rnd(Bsz, T, Cout, seed=12, ints=ints))
    pad = dil * (k - 1) // 2
    xd = x.double().transpose(1, 2).requires_grad_(True)
    y = F.conv1d(xd, w.double(), b.double(), dilation=dil, padding=pad)
    y.backward(dy.double().transpose(1, 2))
    wk = w.permute(0, 2, 1).contiguous().to(DEV)
    out = ops.conv1d(x.to(DEV), wk, b.to(DEV), dilation=dil, out_dtype=torch.float32, kernel=3)
    check(out, y.detach().transpose(1, 2), Cin * k, exact=ints, out_bf16=False)
    dx = torch.empty(Bsz, T, Cin, dtype=torch.float32, device=DEV)
    ops.conv1d_dx(dy.to(DEV), wk, dilation=dil, out=dx, kernel=3)
    check(dx, xd.grad.transpose(1, 2), Cout * k, exact=ints, out_bf16=False)


# ---- fused GEMM + dropout + residual + LayerNorm + PAD zeroing (tts_king_amd/csrc/gemm_ln.hip)
@pytest.mark.parametrize("M,K,seg", [(6768, 256, 423), (6768, 1024, 423), (1024, 256, 64), (1024, 1024, 64), (45, 264, 15), (33, 64, 33)])
def test_gemm_ln_fwd_matches_reference(M, K, seg):
    from tts_king_amd import ops
    D = 256
    a, w, b = bf(rnd(M, K, seed=21) * 0.5), bf(rnd(D, K, seed=22) * K ** -0.5), rnd(D, seed=23) * 0.1
    res = bf(rnd(M, D, seed=24))
    gamma, beta = 1 + 0.1 * rnd(D, seed=25), 0.1 * rnd(D, seed=26)
    nseg = M // seg
    lens = torch.randint(1, seg + 1, (nseg,), generator=torch.Generator().manual_seed(27))
    lens[0] = seg
    z_ref = a.double() @ w.double().t() + b.double() + res.double()
    pad = (torch.arange(seg)[None, :] >= lens[:, None]).reshape(-1)
    ref = F.layer_norm(z_ref, (D,), gamma.double(), beta.double(), 1e-5).masked_fill(pad[:, None], 0.0)
    out, z, mean, rstd = ops.gemm_ln_fwd(a.to(DEV), w.to(DEV), b.to(DEV), res.to(DEV), gamma.to(DEV), beta.to(DEV), lens.to(DEV), seg)
    torch.cuda.synchronize()
    # z: one bf16 rounding of an O(1) value; out: LayerNorm output O(1) rounded to bf16
    assert float((z.float().cpu().double() - z_ref).abs().max()) <= 2 ** -8 * float(z_ref.abs().max()) + 1e-3
    assert float((out.float().cpu().double() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max()) + 2e-3
    torch.testing.assert_close(mean.cpu().double(), z_ref.mean(1), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rstd.cpu().double(), (z_ref.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-3, atol=1e-4)
    assert float(out.float().cpu()[pad].abs().max() if bool(pad.any()) else 0.0) == 0.0
    # the conv-layout weight (256, 1, K) of w_2 is accepted as is
    out2, *_ = ops.gemm_ln_fwd(a.to(DEV), w.view(D, 1, K).to(DEV), b.to(DEV), res.to(DEV), gamma.to(DEV), beta.to(DEV), lens.to(DEV), seg)
    assert torch.equal(out2, out)
    # the window-path kernel (weights as a fragment-major pack, K = 256 / 1024): same fp32 sums in the same order, same row code
    assert ops.win_ln_supported(K, D) == (K in (256, 1024))
    if ops.win_ln_supported(K, D):
        pk = torch.empty(D * K, dtype=torch.bfloat16, device=DEV)
        ops.win_conv_pack_items([(w.view(D, 1, K).to(DEV), pk, False)])
        out3, z3, mean3, rstd3 = ops.win_ln_fwd(a.to(DEV), pk, b.to(DEV), res.to(DEV), gamma.to(DEV), beta.to(DEV), lens.to(DEV), seg)
        assert float((out3.float() - out.float()).abs().max()) <= 2 ** -7 * float(ref.abs().max()) + 2e-3
        assert float((z3.float().cpu().double() - z_ref).abs().max()) <= 2 ** -8 * float(z_ref.abs().max()) + 1e-3
        torch.testing.assert_close(mean3, mean, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(rstd3, rstd, rtol=1e-3, atol=1e-4)
        assert float(out3.float().cpu()[pad].abs().max() if bool(pad.any()) else 0.0) == 0.0


def test_gemm_ln_fwd_dropout_matches_unfused_masks():
    """Same (seed, step, site) -> the fused kernel drops exactly the elements ttsk_layernorm_fwd drops, so ttsk_layernorm_bwd
    regenerates the right mask for either forward."""
    from tts_king_amd import ops
    M, K, D, p = 2048, 256, 256, 0.2
    a, w = bf(rnd(M, K, seed=31)), bf(rnd(D, K, seed=32) * K ** -0.5)
    b = 3.0 + 0.1 * rnd(D, seed=33)                       # keeps |y| away from 0 so that "dropped" is visible as exactly 0
    gamma, beta = torch.ones(D), torch.zeros(D)
    st = ops.optim_state(DEV, seed=77)
    rng = ops.rng_of(st)
    _, z_f, mean_f, rstd_f = ops.gemm_ln_fwd(a.to(DEV), w.to(DEV), b.to(DEV), None, gamma.to(DEV), beta.to(DEV), None, 0, p_pre=p, site_pre=9, rng=rng)
    y = ops.linear(a.to(DEV), w.to(DEV), b.to(DEV))
    _, z_u, mean_u, rstd_u, _ = ops.layernorm_fwd(y, None, gamma.to(DEV), beta.to(DEV), None, 0, p_pre=p, site_pre=9, rng=rng)
    pk = torch.empty(D * K, dtype=torch.bfloat16, device=DEV)
    ops.win_conv_pack_items([(w.view(D, 1, K).to(DEV), pk, False)])
    _, z_w, _, _ = ops.win_ln_fwd(a.to(DEV), pk, b.to(DEV), None, gamma.to(DEV), beta.to(DEV), None, 0, p_pre=p, site_pre=9, rng=rng)
    assert torch.equal(z_w == 0, z_f == 0)                  # the window-path kernel drops the same elements
    zf, zu = z_f.float().cpu(), z_u.float().cpu()
    assert torch.equal(zf == 0, zu == 0)
    keep = zu != 0
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.01
    torch.testing.assert_close(zf[keep], zu[keep], rtol=2 ** -6, atol=1e-3)      # y rounded to bf16 (unfused) vs not (fused)
    yref = (a.double() @ w.double().t() + b.double()) / (1 - p)
    assert float((zf.double()[keep] - yref[keep]).abs().max()) <= 2 ** -8 * float(yref.abs().max()) + 1e-3


@pytest.mark.parametrize("M,N,K,kern,splits", [(6768, 256, 768, 0, 0), (1024, 256, 768, 3, 1), (423, 256, 9216, 2, 4), (1024, 256, 256, 1, 2)])
def test_raw_slabs_sum_to_the_product_and_feed_layernorm_bwd(M, N, K, kern, splits):
    """TTSK_GEMM_RAW_SLABS: the un-reduced split-K tiles sum to A @ B; ttsk_layernorm_bwd_slabs(slabs, R) equals
    ttsk_layernorm_bwd on dout = sum(slabs) + R up to the bf16 rounding of dout it avoids."""
    from tts_king_amd import ops
    a, b = bf(rnd(M, K, seed=41)), bf(rnd(K, N, seed=42) * K ** -0.5)
    sl = ops.gemm(a.to(DEV), b.to(DEV), None, M, N, K, K, N, N, flags=ops.B_TR, kernel=kern, splits=splits, raw=True)
    assert sl.stride == M * N and sl.splits >= 1 and sl.ws.numel() >= sl.splits * M * N
    got = sl.ws[:sl.splits * M * N].view(sl.splits, M, N).sum(0)
    check(got, a.double() @ b.double(), K, out_bf16=False)
    # LayerNorm backward straight from the slabs
    D = N
    r = bf(rnd(M, D, seed=43))
    z = bf(rnd(M, D, seed=44))
    mean, rstd = z.float().mean(1), (z.float().var(1, unbiased=False) + 1e-5).rsqrt()
    gamma, beta = 1 + 0.1 * rnd(D, seed=45), 0.1 * rnd(D, seed=46)
    dout = (got.float() + r.to(DEV).float()).to(torch.bfloat16)
    dz_a, _, part_a, nblk = ops.layernorm_bwd(dout, z.to(DEV), mean.to(DEV), rstd.to(DEV), gamma.to(DEV), beta.to(DEV))
    dz_b, _, part_b, nblk_b = ops.layernorm_bwd(None, z.to(DEV), mean.to(DEV), rstd.to(DEV), gamma.to(DEV), beta.to(DEV), slabs=sl, R=r.to(DEV))
    assert nblk == nblk_b
    err = float((dz_a.float() - dz_b.float()).abs().max())
    assert err <= 2 ** -6 * float(dz_a.float().abs().max()) + 1e-3, err
    torch.testing.assert_close(part_a.sum(0), part_b.sum(0), rtol=2e-2, atol=0.5)


@pytest.mark.parametrize("B,S,Cout,K", [(16, 423, 1024, 9), (16, 64, 1024, 9), (2, 112, 256, 9), (3, 113, 512, 3), (1, 5, 256, 9), (2, 500, 1024, 1)])
def test_ffn_conv_window_kernel(B, S, Cout, K):
    """ttsk_ffn_conv_fwd (window kernel for the FFT block's w_1: SubLayers.py:93-101) against fp64 conv1d on the same bf16 operands
    and against the implicit-GEMM conv it replaces (same operands, another accumulation order), zero padding per utterance, tile
    multiples / ragged / shorter-than-a-tile lengths, with and without ReLU."""
    from tts_king_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(S * 7 + K)
    x = bf(torch.randn(B, S, 256, generator=g)).to(DEV)
    W = bf(torch.randn(Cout, K, 256, generator=g) * (256 * K) ** -0.5).to(DEV)
    bias = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    assert ops.ffn_conv_supported(256, Cout, K) and not ops.ffn_conv_supported(128, Cout, K) and not ops.ffn_conv_supported(256, Cout, 11)
    ref = F.conv1d(x.double().cpu().transpose(1, 2), W.double().cpu().permute(0, 2, 1), bias.double().cpu(), padding=(K - 1) // 2).transpose(1, 2)
    for relu in (True, False):
        got = ops.ffn_conv_fwd(x, W, bias, relu=relu)
        want = ref.clamp(min=0) if relu else ref
        err = float((got.double().cpu() - want).abs().max())
        assert err <= 2 ** -8 * float(want.abs().max()) + 1e-3, (relu, err, float(want.abs().max()))
        old = ops.conv1d(x, W, bias, flags=ops.RELU if relu else 0)
        d = float((got.float() - old.float()).abs().max())
        assert d <= 2 ** -7 * float(want.abs().max()), (relu, d)          # both round the same fp32 sums (up to their order) to bf16
        # fragment-major repack of the weights (what the model feeds the kernel): bit-identical result
        pk = ops.ffn_pack_weight(W)
        assert torch.equal(ops.ffn_conv_fwd(x, W, bias, relu=relu, packed=pk), got)
    pk2 = torch.empty(2 * W.numel(), dtype=torch.bfloat16, device=DEV)
    ops.ffn_pack_weight_batch([W, W], [pk2[:W.numel()], pk2[W.numel():]])
    assert torch.equal(pk2[:W.numel()], pk) and torch.equal(pk2[W.numel():], pk)


@pytest.mark.parametrize("B,S,K", [(16, 423, 5), (2, 64, 5), (3, 65, 3), (1, 7, 5)])
def test_win_conv_postnet_shapes(B, S, K):
    """ttsk_win_conv at Cin = Cout = 512 (PostNet convs, Layers.py:85-129): forward with fp32 output against fp64 and against the
    implicit-GEMM conv; the input gradient as a forward conv on the transposed, tap-flipped pack against conv1d_dx and fp64."""
    from tts_king_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(S * 11 + K)
    x = bf(torch.randn(B, S, 512, generator=g)).to(DEV)
    W = bf(torch.randn(512, K, 512, generator=g) * (512 * K) ** -0.5).to(DEV)
    bias = (0.1 * torch.randn(512, generator=g)).to(DEV)
    pk, pkt = torch.empty(W.numel(), dtype=torch.bfloat16, device=DEV), torch.empty(W.numel(), dtype=torch.bfloat16, device=DEV)
    ops.win_conv_pack_batch([W], [pk])
    ops.win_conv_pack_batch([W], [pkt], transpose=True)
    Wt = W.double().cpu().permute(0, 2, 1)                                  # torch layout (Cout, Cin, k)
    ref = F.conv1d(x.double().cpu().transpose(1, 2), Wt, bias.double().cpu(), padding=(K - 1) // 2).transpose(1, 2)
    got = ops.win_conv(x, pk, 512, K, bias=bias, out_dtype=torch.float32)
    assert got.dtype == torch.float32
    assert float((got.double().cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-4          # fp32 accumulate of bf16 products
    old = ops.conv1d(x, W, bias, out_dtype=torch.float32)
    assert float((got - old).abs().max()) <= 1e-4 * float(ref.abs().max())
    # input gradient: dx = conv_transpose of dy with W = forward conv of dy with the flipped, channel-swapped weights
    dref = F.conv_transpose1d(x.double().cpu().transpose(1, 2), Wt, padding=(K - 1) // 2).transpose(1, 2)
    dgot = ops.win_conv(x, pkt, 512, K)
    assert dgot.dtype == torch.bfloat16
    assert float((dgot.double().cpu() - dref).abs().max()) <= 2 ** -8 * float(dref.abs().max()) + 1e-3
    dold = ops.conv1d_dx(x, W)
    assert float((dgot.float() - dold.float()).abs().max()) <= 2 ** -7 * float(dref.abs().max())


def test_win_conv_k1_projections_gate_and_item_packs():
    """The k = 1 uses of the window kernel in the decoder blocks: q|k|v projection (256 -> 768, bias), fc input gradient (transposed
    pack) and w_2 input gradient (256 -> 1024, transposed pack, zeroed where the saved ReLU output is <= 0: SubLayers.py:96) against
    the GEMM paths; all three packs written by ONE ttsk_win_conv_pack_items launch."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(5)
    B, S, d, Fh = 3, 130, 256, 1024
    x = bf(torch.randn(B, S, d, generator=g)).to(DEV)
    Wq = bf(torch.randn(3 * d, 1, d, generator=g) * d ** -0.5).to(DEV)
    bq = (0.1 * torch.randn(3 * d, generator=g)).to(DEV)
    Wf = bf(torch.randn(d, 1, d, generator=g) * d ** -0.5).to(DEV)
    W2 = bf(torch.randn(d, 1, Fh, generator=g) * Fh ** -0.5).to(DEV)            # storage (Cout = 256, 1, Cin = 1024)
    h = bf(torch.randn(B, S, Fh, generator=g)).clamp(min=0).to(DEV)             # a ReLU output: zeros and positives
    pq, pf, p2 = (torch.empty(w.numel(), dtype=torch.bfloat16, device=DEV) for w in (Wq, Wf, W2))
    ops.win_conv_pack_items([(Wq, pq, False), (Wf, pf, True), (W2, p2, True)])
    qkv = ops.win_conv(x, pq, 3 * d, 1, bias=bq)
    assert torch.equal(qkv.view(-1, 3 * d), ops.linear(x.view(-1, d), Wq.view(3 * d, d), bq))
    do = ops.win_conv(x, pf, d, 1)
    ref = ops.linear_dx(x.view(-1, d), Wf.view(d, d))
    assert float((do.view(-1, d).float() - ref.float()).abs().max()) <= 2 ** -7 * float(ref.float().abs().max())
    # the attention backward's delta written by the fc input-gradient conv: rowsum per 128-column head of (stored dO) * o32
    o32 = torch.randn(B * S, d, generator=g).to(DEV)
    delta = torch.empty(B * 2, S, dtype=torch.float32, device=DEV)
    do2 = ops.win_conv(x, pf, d, 1, delta_o32=o32, delta_out=delta)
    assert torch.equal(do2, do)
    want = (do.view(B, S, 2, 128).float() * o32.view(B, S, 2, 128)).sum(-1).permute(0, 2, 1).reshape(B * 2, S)
    assert float((delta - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-5
    dh = ops.win_conv(x, p2, Fh, 1, gate=h)
    ref = ops.conv1d_dx(x, W2, G=h)
    assert float((dh.float() - ref.float()).abs().max()) <= 2 ** -7 * float(ref.float().abs().max())
    assert torch.equal(dh == 0, (ref == 0)) or float(((dh == 0) != (ref == 0)).float().mean()) < 1e-3
    assert bool((dh[h <= 0] == 0).all())


@pytest.mark.parametrize("rows", [16 * 423, 1024, 37])
def test_qkv_dx_kernel(rows):
    """ttsk_qkv_dx (dqkv x W' + R on 32-row tiles) against the GEMM path it replaces for the first block of a stack and fp64."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(rows)
    d = 256
    dqkv = bf(torch.randn(rows, 3 * d, generator=g)).to(DEV)
    W = bf(torch.randn(3 * d, 1, d, generator=g) * (3 * d) ** -0.5).to(DEV)
    R = bf(torch.randn(rows, d, generator=g)).to(DEV)
    pk = torch.empty(W.numel(), dtype=torch.bfloat16, device=DEV)
    ops.win_conv_pack_items([(W, pk, True)])
    for res in (R, None):
        got = ops.qkv_dx(dqkv, pk, R=res).float().cpu()
        ref = dqkv.double().cpu() @ W.view(3 * d, d).double().cpu() + (res.double().cpu() if res is not None else 0)
        old = ops.linear_dx(dqkv, W.view(3 * d, d), R=res).float().cpu()
        assert float((got.double() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())
        assert float((got - old).abs().max()) <= 2 ** -7 * float(ref.abs().max())


@pytest.mark.parametrize("B,S,limit", [(16, 423, None), (2, 448, 423), (3, 70, 61), (1, 64, None)])
def test_win_conv_emits_batchnorm_partials(B, S, limit):
    """ttsk_win_conv_stats (PostNet 512 -> 512, k = 5): output bit-identical to ttsk_win_conv, and its statistics partials give
    ttsk_bn_train_apply the mean / rstd that ttsk_bn_stats_slab computes from the stored rows (frame limit included)."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(B * 7 + S)
    C = 512
    x = bf(torch.randn(B, S, C, generator=g)).to(DEV)
    W = bf(torch.randn(C, 5, C, generator=g) * (5 * C) ** -0.5).to(DEV)
    bias = (0.1 * torch.randn(C, generator=g)).to(DEV)
    pk = torch.empty(W.numel(), dtype=torch.bfloat16, device=DEV)
    ops.win_conv_pack_items([(W, pk, False)])
    fl = None if limit is None else (torch.tensor([limit], dtype=torch.int32, device=DEV), S)
    want = ops.win_conv(x, pk, C, 5, bias=bias, out_dtype=torch.float32)
    got, stats = ops.win_conv_stats(x, pk, C, 5, bias=bias, frame_limit=fl)
    assert torch.equal(got, want) and stats.shape == (B * ((S + 63) // 64), 2 * C)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    rows = B * S
    z = lambda: (torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV))
    o0, m0, r0 = ops.bn_train(want.view(rows, C), *z(), gamma, beta, True, frame_limit=fl)
    o1, m1, r1 = ops.bn_train(got.view(rows, C), *z(), gamma, beta, True, frame_limit=fl, partials=stats)
    np.testing.assert_allclose(m1.cpu().numpy(), m0.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r1.cpu().numpy(), r0.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert float((o1.float() - o0.float()).abs().max()) <= 2 ** -7


@pytest.mark.parametrize("B,S,limit,p", [(16, 423, None, 0.5), (2, 448, 423, 0.5), (3, 70, 61, 0.0), (1, 64, None, 0.5)])
def test_win_conv_emits_batchnorm_backward_partials(B, S, limit, p):
    """ttsk_win_conv_bnb (a PostNet 512 -> 512 conv's input gradient on its transposed pack, bf16 out): output bit-identical to
    ttsk_win_conv, and its statistics partials give ttsk_bn_bwd_apply_slab the sums that ttsk_bn_bwd_stats_slab computes from the stored
    gradient — so dx, dgamma and dbeta of the layer below agree (tanh, dropout keep bits and the frame limit included)."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(B * 11 + S)
    C = 512
    dy = bf(torch.randn(B, S, C, generator=g)).to(DEV)
    W = bf(torch.randn(C, 5, C, generator=g) * (5 * C) ** -0.5).to(DEV)
    pkt = torch.empty(W.numel(), dtype=torch.bfloat16, device=DEV)
    ops.win_conv_pack_items([(W, pkt, True)])
    fl = None if limit is None else (torch.tensor([limit], dtype=torch.int32, device=DEV), S)
    rows = B * S
    yc = torch.randn(rows, C, generator=g).to(DEV)                      # the layer below: conv output (fp32), statistics, affine, keep bits
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    rng = ops.rng_of(ops.optim_state(DEV, seed=1234)) if p > 0 else None
    z = lambda: (torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV))
    _, mean, rstd, keep = ops.bn_train(yc, *z(), gamma, beta, True, p=p, site=301, rng=rng, frame_limit=fl, want_keep=True)
    want = ops.win_conv(dy, pkt, C, 5)
    got, stats = ops.win_conv_bnb(dy, pkt, C, 5, yc, mean, rstd, gamma, beta, True, p=p, keep=keep, frame_limit=fl)
    assert torch.equal(got, want) and stats.shape == (B * ((S + 63) // 64), 2 * C)
    outs = []
    for part in (None, stats):
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dx = ops.bn_bwd(want.view(rows, C), yc, mean, rstd, gamma, beta, True, p=p, site=301, rng=rng, dgamma=dg, dbeta=db, frame_limit=fl,
                        keep=keep, accumulate=False, partials=part)
        outs.append((dx.float().cpu(), dg.cpu(), db.cpu()))
    (dx0, dg0, db0), (dx1, dg1, db1) = outs
    sc = float(dg0.abs().max()) + float(db0.abs().max())
    print("BN backward from conv partials: max |dgamma diff| %.2e, |dbeta diff| %.2e of %.2e" % (float((dg1 - dg0).abs().max()), float((db1 - db0).abs().max()), sc))
    np.testing.assert_allclose(dg1.numpy(), dg0.numpy(), rtol=2e-5, atol=2e-5 * sc)
    np.testing.assert_allclose(db1.numpy(), db0.numpy(), rtol=2e-5, atol=2e-5 * sc)
    assert float((dx1 - dx0).abs().max()) <= 2 ** -7 * float(dx0.abs().max())


@pytest.mark.parametrize("B,S,K", [(3, 130, 1024), (16, 423, 1024), (2, 33, 256), (16, 64, 1024)])
def test_win_ln_with_qkv_projection_equals_two_launches(B, S, K):
    """ttsk_win_ln_proj_fwd (w_2 / fc + dropout + residual + LayerNorm, then the NEXT block's q|k|v projection of the output rows in
    the same kernel) against ttsk_win_ln_fwd + ttsk_win_conv, bit for bit; ragged lengths (zeroed PAD rows project to the bias)."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(B * 100 + S + K)
    d, rows = 256, B * S
    x = bf(torch.randn(rows, K, generator=g)).to(DEV)
    W = bf(torch.randn(d, 1, K, generator=g) * K ** -0.5).to(DEV)
    Wq = bf(torch.randn(3 * d, 1, d, generator=g) * d ** -0.5).to(DEV)
    bias, bq = (0.1 * torch.randn(d, generator=g)).to(DEV), (0.1 * torch.randn(3 * d, generator=g)).to(DEV)
    res = bf(torch.randn(rows, d, generator=g)).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV), (0.1 * torch.randn(d, generator=g)).to(DEV)
    lens = torch.randint(max(1, S // 2), S + 1, (B,), generator=g).to(DEV)
    pw, pq = (torch.empty(w.numel(), dtype=torch.bfloat16, device=DEV) for w in (W, Wq))
    ops.win_conv_pack_items([(W, pw, False), (Wq, pq, False)])
    rng = ops.rng_of(ops.optim_state(DEV, seed=4))
    o0, z0, m0, r0 = ops.win_ln_fwd(x, pw, bias, res, gamma, beta, lens, S, p_pre=0.1, site_pre=5, rng=rng)
    q0 = ops.win_conv(o0.view(B, S, d), pq, 3 * d, 1, bias=bq).view(rows, 3 * d)
    o1, z1, m1, r1, q1 = ops.win_ln_fwd(x, pw, bias, res, gamma, beta, lens, S, p_pre=0.1, site_pre=5, rng=rng, proj=(pq, bq))
    assert torch.equal(o1, o0) and torch.equal(z1, z0) and torch.equal(m1, m0) and torch.equal(r1, r0)
    assert torch.equal(q1, q0)
    pad = (torch.arange(S, device=DEV)[None, :] >= lens[:, None]).reshape(-1)
    if bool(pad.any()):
        assert torch.equal(q1[pad], bq.to(torch.bfloat16)[None, :].expand(int(pad.sum()), -1))


@pytest.mark.parametrize("B,S,p", [(3, 130, 0.1), (16, 423, 0.1), (2, 33, 0.0), (16, 64, 0.2)])
def test_layernorm_bwd_with_projection_equals_two_launches(B, S, p):
    """ttsk_layernorm_bwd_proj (LayerNorm backward + the k = 1 input-gradient conv on its dy in one kernel) against the two launches it
    replaces, bit for bit: w_2's dX (256 -> 1024, ReLU gate) behind a bf16 upstream gradient, and fc's dX (256 -> 256, attention delta)
    behind split-K slabs + residual; ragged lengths (PAD rows), dropout on and off."""
    from tts_king_amd import ops

    def same(a, b, what):
        bad = (a.float() != b.float()) | (a.float().isnan() != b.float().isnan())
        assert not bool(bad.any()), (what, int(bad.sum()), bad.nonzero()[:4].tolist(), a[bad][:4].tolist(), b[bad][:4].tolist())

    g = torch.Generator().manual_seed(B * 1000 + S)
    d, Fh, rows = 256, 1024, B * S
    z = bf(torch.randn(rows, d, generator=g)).to(DEV)
    mean, rstd = (0.1 * torch.randn(rows, generator=g)).to(DEV), (0.5 + torch.rand(rows, generator=g)).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV), (0.1 * torch.randn(d, generator=g)).to(DEV)
    lens = torch.randint(max(1, S // 2), S + 1, (B,), generator=g).to(DEV)
    dout = bf(torch.randn(rows, d, generator=g)).to(DEV)
    Wf = bf(torch.randn(d, 1, d, generator=g) * d ** -0.5).to(DEV)
    W2 = bf(torch.randn(d, 1, Fh, generator=g) * Fh ** -0.5).to(DEV)
    h = bf(torch.randn(rows, Fh, generator=g)).clamp(min=0).to(DEV)
    pf, p2 = (torch.empty(w.numel(), dtype=torch.bfloat16, device=DEV) for w in (Wf, W2))
    ops.win_conv_pack_items([(Wf, pf, True), (W2, p2, True)])
    rng = ops.rng_of(ops.optim_state(DEV, seed=3))
    # (1) bf16 upstream gradient, gate
    dz0, dy0, part0, n0 = ops.layernorm_bwd(dout, z, mean, rstd, gamma, beta, lens, S, p_pre=p, site_pre=7, rng=rng)
    dh0 = ops.win_conv(dy0.view(B, S, d), p2, Fh, 1, gate=h.view(B, S, Fh))
    dz1, dy1, part1, n1, dh1 = ops.layernorm_bwd_proj(dout, z, mean, rstd, gamma, p2, Fh, lens, S, p_pre=p, site_pre=7, rng=rng, gate=h)
    same(dz1, dz0, "dz"), same(dy1, dy0, "dy"), same(dh1, dh0.view(rows, Fh), "dh")
    assert n1 == (rows + 31) // 32
    s0, s1 = part0[:n0].double().sum(0), part1.double().sum(0)
    assert float((s0 - s1).abs().max()) <= 1e-5 * float(s0.abs().max())
    # (2) split-K slabs + residual, delta
    nsp = 3
    ws = torch.randn(nsp, rows, d, generator=g).to(DEV).contiguous()
    sl = ops.Slabs(ws, nsp, rows * d)
    R = bf(torch.randn(rows, d, generator=g)).to(DEV)
    o32 = torch.randn(rows, d, generator=g).to(DEV)
    dz0, dy0, part0, n0 = ops.layernorm_bwd(None, z, mean, rstd, gamma, beta, lens, S, p_pre=p, site_pre=9, rng=rng, slabs=sl, R=R)
    del0 = torch.empty(B * 2, S, dtype=torch.float32, device=DEV)
    do0 = ops.win_conv(dy0.view(B, S, d), pf, d, 1, delta_o32=o32, delta_out=del0)
    del1 = torch.empty(B * 2, S, dtype=torch.float32, device=DEV)
    dz1, dy1, part1, n1, do1 = ops.layernorm_bwd_proj(None, z, mean, rstd, gamma, pf, d, lens, S, p_pre=p, site_pre=9, rng=rng, slabs=sl, R=R,
                                                      delta_o32=o32, delta_out=del1)
    same(dz1, dz0, "dz (slabs)"), same(dy1, dy0, "dy (slabs)"), same(do1, do0.view(rows, d), "do"), same(del1, del0, "delta")
    s0, s1 = part0[:n0].double().sum(0), part1.double().sum(0)
    assert float((s0 - s1).abs().max()) <= 1e-5 * float(s0.abs().max())
    # (3) the upstream gradient is the q|k|v input gradient of the block behind, computed inside the kernel (dqkv x W' + R): against the
    # split-K slab path (same products, fp32 sums in another order: agreement to bf16 rounding, rare one-ulp differences)
    Wq = bf(torch.randn(3 * d, 1, d, generator=g) * (3 * d) ** -0.5).to(DEV)
    pq = torch.empty(Wq.numel(), dtype=torch.bfloat16, device=DEV)
    ops.win_conv_pack_items([(Wq, pq, True)])
    dqkv = bf(torch.randn(rows, 3 * d, generator=g)).to(DEV)
    slq = ops.win_conv_split(dqkv.view(B, S, 3 * d), pq, d, 1)
    dz0, dy0, part0, n0, dh0 = ops.layernorm_bwd_proj(None, z, mean, rstd, gamma, p2, Fh, lens, S, p_pre=p, site_pre=11, rng=rng, slabs=slq, R=R, gate=h)
    dz1, dy1, part1, n1, dh1 = ops.layernorm_bwd_proj(None, z, mean, rstd, gamma, p2, Fh, lens, S, p_pre=p, site_pre=11, rng=rng, R=R, gate=h,
                                                      pre=(dqkv, pq))
    for got, want, what in ((dz1, dz0, "dz"), (dy1, dy0, "dy")):
        gf, wf = got.float(), want.float()
        tol = 2 ** -7 * wf.abs() + 2 ** -9 * float(wf.abs().mean())
        assert bool(((gf - wf).abs() <= tol).all()), (what, float((gf - wf).abs().max()))
        assert float((gf != wf).float().mean()) < 0.02, what
    # dh = dy x W2': a sum of 256 products, each dy possibly one bf16 ulp apart
    assert float((dh1.float() - dh0.float()).abs().max()) <= 2 ** -6 * float(dh0.float().abs().max())
    assert torch.equal(dh1 == 0, dh0 == 0) or float(((dh1 == 0) != (dh0 == 0)).float().mean()) < 1e-3       # the ReLU gate's zeros
    s0, s1 = part0.double().sum(0), part1.double().sum(0)
    assert float((s0 - s1).abs().max()) <= 1e-4 * float(s0.abs().max())


@pytest.mark.parametrize("B,S,Cin,K", [(16, 423, 1024, 9), (16, 64, 1024, 9), (2, 130, 768, 1), (1, 9, 512, 3)])
def test_win_conv_split_slabs(B, S, Cin, K):
    """ttsk_win_conv_split: the input gradient of a conv with a wide contraction (w_1: 1024 channels x 9 taps; q|k|v: 768) as 256-channel
    slices into fp32 slabs; their sum against conv1d_dx's fp32 result and fp64, and through ttsk_layernorm_bwd_slabs' consumer path."""
    from tts_king_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(S + Cin + K)
    dy = bf(torch.randn(B, S, Cin, generator=g)).to(DEV)
    W = bf(torch.randn(Cin, K, 256, generator=g) * (Cin * K) ** -0.5).to(DEV)          # storage of the forward conv 256 -> Cin
    pk = torch.empty(W.numel(), dtype=torch.bfloat16, device=DEV)
    ops.win_conv_pack_items([(W, pk, True)])
    sl = ops.win_conv_split(dy, pk, 256, K)
    assert sl.splits == Cin // 256 and sl.stride == B * S * 256
    got = sl.ws.view(sl.splits, B * S, 256).double().sum(0).cpu()
    ref = F.conv_transpose1d(dy.double().cpu().transpose(1, 2), W.double().cpu().permute(0, 2, 1), padding=(K - 1) // 2).transpose(1, 2).reshape(B * S, 256)
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-4
    old = ops.conv1d_dx(dy, W, out_dtype=torch.float32) if False else ops.conv1d_dx(dy, W)
    assert float((got.float() - old.view(B * S, 256).float().cpu()).abs().max()) <= 2 ** -7 * float(ref.abs().max())


@pytest.mark.parametrize("Bn,S,Cout,Cin,lens", [(16, 423, 1024, 256, "ragged"), (16, 423, 1024, 256, None), (3, 40, 256, 32, "ragged"),
                                                (2, 7, 256, 64, None), (5, 64, 512, 256, "zeros"), (1, 33, 256, 32, None)])
def test_dwconv_vs_fp64_and_grouped_gemm(Bn, S, Cout, Cin, lens):
    """ttsk_dwconv_batch (csrc/dwconv.hip): the k = 9 Conv1d weight gradient with the taps sharing one dY fetch and one X window —
    against fp64 on the same bf16 operands (what torch's conv backward gives for `weight.grad`, SubLayers.py:96), against the grouped
    GEMM path it replaces (ops.conv1d_dw), with `lens` (rows past an utterance's length carry no gradient and are skipped), with
    overwrite and accumulate, at the step's shape, at shapes shorter than one K step, and with empty utterances."""
    from tts_king_amd import ops
    k = 9
    g = torch.Generator().manual_seed(Bn * S + Cin)
    dy = torch.randn(Bn, S, Cout, generator=g).to(torch.bfloat16)
    x = torch.randn(Bn, S, Cin, generator=g).to(torch.bfloat16)
    ln = None
    if lens == "ragged":
        ln = torch.randint(max(1, S // 2), S + 1, (Bn,), generator=g)
        ln[0] = S
    elif lens == "zeros":
        ln = torch.tensor([S, 0, 17, 0, 1][:Bn])
    if ln is not None:                                   # PAD rows: zero gradient (and zero activations, as in the FFT blocks)
        m = torch.arange(S)[None, :] >= ln[:, None]
        dy[m] = 0
        x[m] = 0
    # fp64: dW[co, j, ci] = sum_b sum_t dy[b, t, co] * x[b, t + j - 4, ci]
    xp = torch.nn.functional.pad(x.double(), (0, 0, k // 2, k // 2))
    want = torch.stack([torch.einsum("btc,bti->ci", dy.double(), xp[:, j:j + S]) for j in range(k)], dim=1)
    dyd, xd = dy.to(DEV), x.to(DEV)
    lnd = ln.to(DEV) if ln is not None else None
    assert ops.dwconv_supported(Cout, Cin, k)
    out = torch.full((Cout, k, Cin), 3.0, device=DEV)
    ops.dwconv_batch([(dyd, xd, out, lnd, False)])
    scale = float(want.abs().max())
    err = float((out.double().cpu() - want).abs().max()) / scale
    print("dwconv B=%d S=%d %dx%d lens=%s: max err %.2e of max |dW|" % (Bn, S, Cout, Cin, lens, err))
    assert err <= 2e-5                                   # fp32 accumulation of exact bf16 products
    # accumulate on top of itself, and two problems in one launch
    out2 = out.clone()
    other = torch.zeros_like(out)
    ops.dwconv_batch([(dyd, xd, out2, lnd, True), (dyd, xd, other, None, False)])
    assert float((out2 - 2 * out).abs().max()) <= 1e-5 * scale
    assert float((other.double().cpu() - want).abs().max()) / scale <= 2e-5     # lens = None walks the zero rows: same sum
    # the grouped GEMM path on the same operands
    ref = torch.zeros_like(out)
    ops.conv1d_dw(dyd, xd, ref, k=k, accumulate=False)
    assert float((ref - out).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("Bn,S,Cout,Cin,k,lens,splits", [(16, 423, 256, 1024, 1, "ragged", 4), (16, 423, 768, 256, 1, "ragged", 1),
                                                         (16, 423, 512, 512, 5, None, 2), (3, 40, 256, 256, 1, "ragged", 3),
                                                         (2, 7, 256, 256, 3, None, 1), (5, 64, 256, 512, 1, "zeros", 5),
                                                         (1, 33, 512, 256, 1, None, 1), (7, 100, 256, 256, 5, "ragged", 2)])
def test_dwgemm_vs_fp64(Bn, S, Cout, Cin, k, lens, splits):
    """ttsk_dwgemm_batch (csrc/dwgemm.hip): weight gradients with Cout, Cin multiples of 256 on the 256x256-tile kernel — against fp64
    on the same bf16 operands (torch's `weight.grad` of a Linear / Conv1d: SubLayers.py:41-43,62,97, Layers.py:85-129), unsplit and
    split over utterance ranges (slabs + ttsk_gemm_reduce_batch), overwrite and accumulate, with `lens`, with taps, with empty
    utterances, with fewer rows than one K step."""
    from tts_king_amd import ops
    from tts_king_amd import lib as L
    import ctypes as C
    g = torch.Generator().manual_seed(Bn * S + Cin + k)
    dy = torch.randn(Bn, S, Cout, generator=g).to(torch.bfloat16)
    x = torch.randn(Bn, S, Cin, generator=g).to(torch.bfloat16)
    ln = None
    if lens == "ragged":
        ln = torch.randint(max(1, S // 2), S + 1, (Bn,), generator=g)
        ln[0] = S
    elif lens == "zeros":
        ln = torch.tensor([S, 0, 17, 0, 1][:Bn])
    if ln is not None:
        m = torch.arange(S)[None, :] >= ln[:, None]
        dy[m] = 0
        x[m] = 0
    xp = torch.nn.functional.pad(x.double(), (0, 0, k // 2, k // 2))
    want = torch.stack([torch.einsum("btc,bti->ci", dy.double(), xp[:, j:j + S]) for j in range(k)], dim=1)
    scale = float(want.abs().max())
    dyd, xd = dy.to(DEV), x.to(DEV)
    lnd = ln.to(DEV) if ln is not None else None
    assert ops.dwgemm_supported(Cout, Cin, k)

    def run(dst, accumulate, sp):
        red = ops.dwgemm_batch([(dyd, xd, dst, lnd, accumulate, sp)])
        if red:
            arr = (L.ReduceItem * len(red))(*[r for r, _ in red])
            L.check(L.load().ttsk_gemm_reduce_batch(arr, len(red), torch.cuda.current_stream().cuda_stream), "reduce")
        torch.cuda.synchronize()
    out = torch.full((Cout, k, Cin), 3.0, device=DEV)
    run(out, False, splits)
    err = float((out.double().cpu() - want).abs().max()) / scale
    print("dwgemm B=%d S=%d %dx%d k=%d lens=%s splits=%d: max err %.2e of max |dW|" % (Bn, S, Cout, Cin, k, lens, splits, err))
    assert err <= 2e-5
    out1 = torch.full((Cout, k, Cin), -1.0, device=DEV)
    run(out1, False, 1)
    assert float((out1 - out).abs().max()) <= 2e-5 * scale          # split or not: the same sum
    run(out1, True, splits)
    assert float((out1 - 2 * out).abs().max()) <= 4e-5 * scale
