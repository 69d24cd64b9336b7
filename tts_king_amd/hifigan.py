"""HiFi-GAN V1 generator inference on MI355X: the reference's `Generator(h)` surface (constructor, weight-normed
`state_dict` keys, `remove_weight_norm()`, `forward((B,80,T)) -> (B,1,T*prod(upsample_rates))`) over hand-written
gfx950 kernels.

reference: hifi/models.py:146-210 (Generator), :12-95 (ResBlock1), :98-143 (ResBlock2), hifiapi.py:11-52.

Activations are bf16 channels-last [B*len][C] (each conv is an implicit GEMM whose contraction index — the input
channel — is contiguous in memory), accumulation is fp32, weights are folded (weight-norm removed), repacked tap-major
and cast to bf16 once.  Forward only: the reference's generator is used for inference (hifiapi.py:32-33 `train`
raises).
"""
import os
import torch
import torch.nn as nn

from . import ops
from . import switches
from .ops import bf16, f16

LRELU_SLOPE = 0.1          # reference: hifi/models.py:9


def get_padding(kernel_size, dilation=1):
    """reference: hifi/vocoder/utils.py:36-37."""
    return int((kernel_size * dilation - dilation) / 2)


class _WNConv(nn.Module):
    """Parameter holder for one weight-normed Conv1d / ConvTranspose1d with the reference's key names:
    `weight_g`, `weight_v`, `bias` before `remove_weight_norm()`, `weight`, `bias` after."""

    def __init__(self, shape, n_bias):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(n_bias), requires_grad=False)          # reference key order: bias, weight_g, weight_v
        self.weight_g = nn.Parameter(torch.ones(shape[0], 1, 1), requires_grad=False)
        self.weight_v = nn.Parameter(torch.zeros(shape), requires_grad=False)

    def fold(self):
        if "weight_g" not in self._parameters:
            return
        g, v = self.weight_g, self.weight_v
        if v.is_cuda:
            w = ops.weight_norm_fold(v.data, g.data.view(-1))
        else:   # folding before .to(device): plain tensor math on the host (load-time only, not the hot path)
            w = v.data * (g.data / v.data.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1))
        del self._parameters["weight_g"], self._parameters["weight_v"]
        self.weight = nn.Parameter(w, requires_grad=False)

    def folded_weight(self):
        if "weight" in self._parameters:
            return self.weight.data
        if not self.weight_v.is_cuda:
            raise ops.L.TtskError("HiFi-GAN generator needs its weights on a HIP device")
        return ops.weight_norm_fold(self.weight_v.data, self.weight_g.data.view(-1))


class _ResBlock(nn.Module):
    """ResBlock1 (three (dilated, plain) conv pairs) or ResBlock2 (two dilated convs).
    reference: hifi/models.py:12-95, :98-143."""

    def __init__(self, channels, kernel_size, dilation, kind):
        super().__init__()
        self.kind, self.k, self.dilation = kind, kernel_size, tuple(dilation)
        mk = lambda: _WNConv((channels, channels, kernel_size), channels)
        if kind == "1":
            self.convs1 = nn.ModuleList([mk() for _ in self.dilation])
            self.convs2 = nn.ModuleList([mk() for _ in self.dilation])
        else:
            self.convs = nn.ModuleList([mk() for _ in self.dilation])

    def all_convs(self):
        return (list(self.convs1) + list(self.convs2)) if self.kind == "1" else list(self.convs)

    def remove_weight_norm(self):
        for c in self.all_convs():
            c.fold()


class Generator(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.h = h
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        c0 = h.upsample_initial_channel
        self.conv_pre = _WNConv((c0, 80, 7), c0)                       # 80 is hard-coded in the reference (:152)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            self.ups.append(_WNConv((c0 // (2 ** i), c0 // (2 ** (i + 1)), k), c0 // (2 ** (i + 1))))
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(self.num_upsamples):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(_ResBlock(ch, k, d, str(h.resblock)))
        self.conv_post = _WNConv((1, ch, 7), 1)
        self._packed = None
        self._packed_key = None
        self.act_dtype = f16         # storage type of activations and packed weights (fp32 accumulate); bf16 also works
        self.window_conv = True      # window-conv kernel at the C = 128 stage; False = implicit-GEMM convs
        self.conv_pair = True        # each (c1, c2) pair of a ResBlock1 as one launch (ttsk_hifi_conv_pair) at C = 128 ...
        self.conv_pair_small = True  # ... and at C = 64 / 32, instead of the six-conv fused kernel
        self.pair_ws = True          # the pairs on the weights-stationary persistent kernel (csrc/pairws.hip, round 6): C = 64 every kernel size, C = 128 k = 3
        self.pair_ws_min_tiles = 1024    # ... for stage tensors of at least this many of its tiles (192 frames at C = 64, 96 at C = 128)
        self.fused = True            # fused ResBlock1 kernel where an instance exists (C in {32,64}); False = conv-by-conv
        self.stream_upsample = True  # stride-2 upsamplers (128->64, 64->32) on the streaming kernel; False = polyphase implicit GEMMs
        self.window_upsample = switches.get("TTSK_HIFI_UPS8") != "0"   # stride-8 upsamplers and 128 -> 64 on the window-conv kernel (fp16); 0 = polyphase GEMMs / streaming kernel
        self.window_conv_pre = True       # conv_pre on the window-conv kernel (False: the implicit GEMM)
        self.loop_upsample = True         # the 256 -> 128 upsampler on ups_loop_kernel (False: win_conv_kernel, one channel group per workgroup)
        self.group_resblocks = True  # conv m of the three MRF ResBlocks as one grouped launch where they run conv by conv (C = 256)
        self.mrf_fused = True        # the last stage (C = 32: three ResBlock1s + average + LeakyReLU + conv_post + tanh) as ONE launch (csrc/mrf32.hip)

    # ------------------------------------------------------------------ reference surface
    def remove_weight_norm(self):
        """reference: hifi/models.py:203-210 — fold g*v/||v|| into `weight`; keys lose `_g/_v`."""
        for l in self.ups:
            l.fold()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.fold()
        self.conv_post.fold()
        self._packed = None

    def reset_parameters(self, seed=1234):
        """Seeded random init (weights_path: null): tts_king_amd.synthetic.seeded_fill statistics."""
        from .synthetic import seeded_fill
        sd = self.state_dict()
        cpu = {k: torch.zeros(v.shape, dtype=v.dtype) for k, v in sd.items()}
        seeded_fill(cpu, seed)
        with torch.no_grad():
            for k, v in sd.items():
                v.copy_(cpu[k])
        self._packed = None

    def _all_convs(self):
        out = [self.conv_pre] + list(self.ups)
        for rb in self.resblocks:
            out += rb.all_convs()
        return out + [self.conv_post]

    def _weights_key(self):
        return tuple((id(p), p._version) for c in self._all_convs() for p in c._parameters.values())

    def _prepare(self):
        """16-bit copies of the folded weights in kernel layouts (tap-major for the implicit-GEMM convs, MFMA-fragment-
        major for the fused ResBlock kernel), rebuilt when any parameter was written."""
        dt = self.act_dtype
        key = self._weights_key() + (dt,)
        if self._packed is not None and self._packed_key == key:
            return self._packed
        pk = {}
        pk["pre"] = (ops.pack_conv_weight(self.conv_pre.folded_weight(), dtype=dt), self.conv_pre.bias.data)
        cpre, kpre = pk["pre"][0].shape[0], pk["pre"][0].shape[1]
        pk["pre_win"] = (ops.hifi_conv_pre_win_pack(pk["pre"][0]) if (self.window_upsample and self.window_conv_pre and dt == torch.float16 and
                                                                      ops.hifi_conv_pre_win_supported(int(pk["pre"][0].shape[2]), cpre, kpre)) else None)
        pk["ups"] = [(ops.pack_conv_weight(u.folded_weight(), transposed=True, dtype=dt), u.bias.data) for u in self.ups]
        # the stride-8 upsamplers and the 128 -> 64 stride-2 one on the window-conv kernel (fp16 rows): a two-tap conv over the input frames
        # with stride * Cout phase-major channels
        pk["ups8"] = [ops.hifi_upsample_win_pack(w, b, uu) if (self.window_upsample and dt == torch.float16 and
                                                               ops.hifi_upsample_win_supported(w.shape[2], w.shape[1], uu, kk)) else None
                      for (w, b), uu, kk in zip(pk["ups"], self.h.upsample_rates, self.h.upsample_kernel_sizes)]
        pk["rb"], pk["rbf"], pk["rbw"] = [], [], []
        for rb in self.resblocks:
            convs = rb.all_convs()
            ch = convs[0].bias.shape[0]
            if rb.kind == "1" and ops.hifi_resblock1_supported(ch, rb.k):
                n = len(rb.dilation)
                order = [convs[m // 2 + (n if m % 2 else 0)] for m in range(2 * n)]          # c1_0, c2_0, c1_1, c2_1, ...
                pk["rbf"].append(([ops.pack_resblock_weight(c.folded_weight(), dtype=dt) for c in order], [c.bias.data for c in order]))
            else:
                pk["rbf"].append(None)
            if rb.kind == "1" and pk["rbf"][-1] is None and (all(ops.hifi_conv_window_supported(ch, rb.k, dd) for dd in rb.dilation) or
                                                             all(ops.hifi_conv_pair_supported(ch, rb.k, dd) for dd in rb.dilation)):
                pk["rbw"].append([ops.pack_resblock_weight(c.folded_weight(), dtype=dt) for c in convs])     # window-conv packs
            else:
                pk["rbw"].append(None)
            pk["rb"].append([(ops.pack_conv_weight(c.folded_weight(), dtype=dt), c.bias.data) for c in convs])
        pk["post"] = (ops.pack_conv_weight(self.conv_post.folded_weight(), dtype=dt), self.conv_post.bias.data)
        self._packed, self._packed_key = pk, key
        return pk

    # ------------------------------------------------------------------ forward
    def _resblock(self, rb, packed, x, xl, wpacks=None):
        """reference: hifi/models.py:88-95 (ResBlock1) / :136-140 (ResBlock2), conv by conv on the implicit-GEMM kernel.
        x = block input, xl = lrelu(x).  Every LeakyReLU is applied by the PRODUCING conv's epilogue (LRELU_OUT, or a
        second output C2 = lrelu(v) next to the raw v the residual path needs), the residual add is an epilogue too."""
        nd = len(rb.dilation)
        for m, d in enumerate(rb.dilation):
            lastp = m == nd - 1
            xl_next = None if lastp else torch.empty_like(x)
            if rb.kind == "1" and wpacks is not None and self.window_conv:
                # C = 128: one window-conv launch per conv (activation window in LDS, weights streamed)
                tl = ops.hifi_conv_window(xl, wpacks[m], packed[m][1], rb.k, d, lrelu_out=True, slope=LRELU_SLOPE)
                x = ops.hifi_conv_window(tl, wpacks[nd + m], packed[nd + m][1], rb.k, 1, R=x, out2=xl_next, slope=LRELU_SLOPE)
            elif rb.kind == "1":
                w1, b1 = packed[m]
                w2, b2 = packed[nd + m]
                tl = ops.conv1d(xl, w1, b1, dilation=d, flags=ops.LRELU_OUT, out_slope=LRELU_SLOPE)
                x = ops.conv1d(tl, w2, b2, R=x, C2=xl_next, flags=0 if lastp else ops.C2_LRELU, out_slope=LRELU_SLOPE)
            else:
                w, b = packed[m]
                x = ops.conv1d(xl, w, b, dilation=d, R=x, C2=xl_next, flags=0 if lastp else ops.C2_LRELU, out_slope=LRELU_SLOPE)
            xl = xl_next
        return x

    def _pair_blocks(self, rbs, packs, a, nxt_slope, fused_for=()):
        """The stage's ResBlock1s on the pair kernel (hifi/models.py:88-95, :190-197): per block three launches, each one
        (c1 dilated -> lrelu -> c2 -> + x); every block's last launch adds into `out` and the last block's scales by 1/num_kernels and
        applies the consumer's LeakyReLU (the MRF average).  packs[j] = (weight packs, biases), both in the order c1_0, c2_0, c1_1,
        c2_1, ...; blocks whose kernel size is in `fused_for` run on the six-conv fused kernel instead (same packs, same modes)."""
        nk = len(rbs)
        out = torch.empty_like(a)
        for j, rb in enumerate(rbs):
            ws, bs = packs[j]
            mode = 0 if j == 0 else (2 if j == nk - 1 else 1)
            fs = nxt_slope if j == nk - 1 else 1.0
            # the persistent kernel walks runs of 192-frame tiles (96 at C = 128), one workgroup per CU, and pays a pipeline fill and drain of one tile each per
            # launch: worth it from ~4 tiles per CU on (the bench batch: 8); one utterance of 5 s is 299 tiles at the last stage and stays on the round-5 kernels
            tt = ops.hifi_conv_pair_ws_tile(a.shape[2])
            ws_kernel = (self.pair_ws and a.is_contiguous() and a.shape[0] * ((a.shape[1] + tt - 1) // tt) >= self.pair_ws_min_tiles and
                         all(ops.hifi_conv_pair_ws_supported(a.shape[2], rb.k, d, a.shape[1]) for d in rb.dilation))
            if rb.k in fused_for and not ws_kernel:
                ops.hifi_resblock1(a, ws, bs, rb.dilation, out, rb.k, mode=mode, scale=1.0 / nk, slope=LRELU_SLOPE, final_slope=fs)
                continue
            x, nd = a, len(rb.dilation)
            for m, d in enumerate(rb.dilation):
                kw = dict(out=out, mode=mode, scale=1.0 / nk, final_slope=fs) if m == nd - 1 else {}
                x = ops.hifi_conv_pair(x, ws[2 * m], bs[2 * m], ws[2 * m + 1], bs[2 * m + 1], rb.k, d, slope=LRELU_SLOPE, ws=ws_kernel, **kw)
        return out

    def _pair_packs(self, pk, i, nk, rbs, C):
        """Weight packs / biases of stage i in `_pair_blocks` order, or None when the pair kernel does not cover the stage."""
        if nk < 2 or not all(rb.kind == "1" and all(ops.hifi_conv_pair_supported(C, rb.k, d) for d in rb.dilation) for rb in rbs):
            return None
        packs = []
        for j, rb in enumerate(rbs):
            n = len(rb.dilation)
            if pk["rbf"][i * nk + j] is not None:                      # packed for the fused kernel: already in pair order
                packs.append(pk["rbf"][i * nk + j])
            elif pk["rbw"][i * nk + j] is not None:                    # window packs: convs1 then convs2
                w, b = pk["rbw"][i * nk + j], [p[1] for p in pk["rb"][i * nk + j]]
                order = [m // 2 + (n if m % 2 else 0) for m in range(2 * n)]
                packs.append(([w[o] for o in order], [b[o] for o in order]))
            else:
                return None
        return packs

    def _resblocks_lockstep(self, rbs, packs, x, xl):
        """The stage's ResBlock1s (one per MRF kernel size, hifi/models.py:190-196) advanced conv by conv TOGETHER: they read
        the same stage input and never each other's outputs, so conv m of every block is one grouped launch (three 192-
        workgroup problems at the C = 256 stage -> one 576-workgroup grid).  Same epilogue fusions as `_resblock`."""
        nd = len(rbs[0].dilation)
        xs, xls = [x] * len(rbs), [xl] * len(rbs)
        for m in range(nd):
            lastp = m == nd - 1
            g1, tls = ops.GemmGroup(), []
            for rb, pk, xlj in zip(rbs, packs, xls):
                tls.append(ops.conv1d(xlj, pk[m][0], pk[m][1], dilation=rb.dilation[m], flags=ops.LRELU_OUT, out_slope=LRELU_SLOPE, group=g1))
            g1.flush()
            g2, nxt = ops.GemmGroup(), []
            for j, (rb, pk) in enumerate(zip(rbs, packs)):
                xl_next = None if lastp else torch.empty_like(x)
                xs[j] = ops.conv1d(tls[j], pk[nd + m][0], pk[nd + m][1], R=xs[j], C2=xl_next, flags=0 if lastp else ops.C2_LRELU,
                                   out_slope=LRELU_SLOPE, group=g2)
                nxt.append(xl_next)
            g2.flush()
            xls = nxt
        return xs

    _stage_marks = None     # bench.py's per-stage timing: a list to which forward appends (name, HIP event) at stage boundaries

    def _mark(self, name):
        if self._stage_marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._stage_marks.append((name, ev))

    def forward(self, x):
        """x (B, 80, T) mel on a HIP device -> (B, 1, T*prod(upsample_rates)) fp32.  reference: hifi/models.py:185-201.

        Dataflow: `al` always holds LeakyReLU(stage output) — the only thing the next ConvTranspose1d / conv_post reads
        (hifi/models.py:188,197) — produced by conv_pre's epilogue, then by the MRF average of each stage."""
        if not x.is_cuda:
            raise ops.L.TtskError("HiFi-GAN Generator.forward needs a HIP device tensor (gpu: 'cuda:0'); there is no CPU path")
        pk = self._prepare()
        h = self.h
        nk = self.num_kernels
        with torch.no_grad():
            self._mark("start")
            a0 = ops.nct_to_ntc(x.float(), self.act_dtype)                                 # (B, T, 80) 16-bit
            if pk["pre_win"] is not None:
                al = ops.hifi_conv_pre_win(a0, pk["pre_win"], pk["pre"][1], pk["pre"][0].shape[0], pk["pre"][0].shape[1], LRELU_SLOPE)   # window kernel
            else:
                al = ops.conv1d(a0, pk["pre"][0], pk["pre"][1], flags=ops.LRELU_OUT, out_slope=LRELU_SLOPE)   # lrelu(conv_pre(x))
            for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
                self._mark("conv_pre" if i == 0 else "mrf%d" % (i - 1))
                wu, bu = pk["ups"][i]
                # slope of the activation that consumes this stage's output: 0.1 before the next upsampler,
                # F.leaky_relu's default 0.01 before conv_post (hifi/models.py:197)
                nxt_slope = LRELU_SLOPE if i + 1 < self.num_upsamples else 0.01
                rbs = [self.resblocks[i * nk + j] for j in range(nk)]
                C_out = wu.shape[1]
                # C = 32: the fused kernel is as fast or faster for every kernel size (86 / 143 / 181 us against 102 / 139 / 178)
                want_pair = (self.conv_pair and self.window_conv) if C_out >= 128 else (self.conv_pair_small and self.fused and C_out == 64)
                ppacks = self._pair_packs(pk, i, nk, rbs, C_out) if want_pair else None
                if ppacks is not None:
                    if pk["ups8"][i] is not None and al.is_contiguous():
                        if self.loop_upsample and ops.hifi_upsample_loop_supported(al.shape[2], C_out, u, k):
                            a = ops.hifi_upsample_loop(al, pk["ups8"][i][0], pk["ups8"][i][1], C_out, u)      # 256 -> 128: channel groups looped per frame tile
                        else:
                            a = ops.hifi_upsample_win(al, pk["ups8"][i][0], pk["ups8"][i][1], C_out, u)
                    elif self.stream_upsample and ops.hifi_upsample2_supported(wu.shape[2], wu.shape[1], u, k) and al.is_contiguous():
                        a = ops.hifi_upsample2(al, wu, bu)
                    else:
                        a = ops.conv_transpose1d(al, wu, bu, u, k)                         # raw x: the pair kernels activate it themselves
                    # measured per block at the bench shape (tools/debug/convpair_micro.py): three pair launches beat the six-conv
                    # fused kernel at C = 64 for k = 7, 11 (228 vs 242 us, 290 vs 370 us) and lose at k = 3 (158 vs 141 us: the block is
                    # then bound by its HBM passes, and the fused kernel makes one instead of three)
                    fused_for = (3,) if C_out == 64 and all(pk["rbf"][i * nk + j] is not None for j in range(nk)) else ()
                    self._mark("ups%d" % i)
                    al = self._pair_blocks(rbs, ppacks, a, nxt_slope, fused_for)
                    continue
                fused = self.fused and all(pk["rbf"][i * nk + j] is not None for j in range(nk)) and nk >= 2
                if fused:
                    if self.stream_upsample and ops.hifi_upsample2_supported(wu.shape[2], wu.shape[1], u, k) and al.is_contiguous():
                        a = ops.hifi_upsample2(al, wu, bu)                                 # both phases from one read of `al`
                    else:
                        a = ops.conv_transpose1d(al, wu, bu, u, k)                         # raw x: the fused blocks activate it themselves
                    self._mark("ups%d" % i)
                    if (self.mrf_fused and i + 1 == self.num_upsamples and nk == 3 and a.is_contiguous() and
                            ops.hifi_mrf32_post_supported(C_out, [rb.k for rb in rbs], pk["post"][0].shape[1])):
                        # the whole last stage — three blocks, their average, LeakyReLU(0.01), conv_post, tanh — in one launch: x is
                        # read once, the waveform written once (hifi/models.py:190-199)
                        ws = [w for j in range(nk) for w in pk["rbf"][i * nk + j][0]]
                        bs = [b for j in range(nk) for b in pk["rbf"][i * nk + j][1]]
                        y = ops.hifi_mrf32_post(a, ws, bs, [rb.dilation for rb in rbs], [rb.k for rb in rbs], pk["post"][0], pk["post"][1],
                                                slope=LRELU_SLOPE, final_slope=nxt_slope, scale=1.0 / nk)
                        self._mark("mrf%d" % i)
                        self._mark("conv_post")
                        return y
                    out = torch.empty_like(a)
                    for j, rb in enumerate(rbs):
                        ws, bs = pk["rbf"][i * nk + j]
                        lastb = j == nk - 1
                        ops.hifi_resblock1(a, ws, bs, rb.dilation, out, rb.k, mode=0 if j == 0 else (2 if lastb else 1),
                                           scale=1.0 / nk, slope=LRELU_SLOPE, final_slope=nxt_slope if lastb else 1.0)
                    al = out
                    continue
                windowed = self.window_conv and all(pk["rbw"][i * nk + j] is not None and ops.hifi_conv_window_supported(C_out, rbs[j].k, dd)
                                                    for j in range(nk) for dd in rbs[j].dilation)
                axl = torch.empty(al.shape[0], al.shape[1] * u, wu.shape[1], dtype=al.dtype, device=al.device)
                a = ops.conv_transpose1d(al, wu, bu, u, k, C2=axl, flags=ops.C2_LRELU, out_slope=LRELU_SLOPE)   # x and lrelu(x)
                self._mark("ups%d" % i)
                if self.group_resblocks and all(rb.kind == "1" for rb in rbs) and len({len(rb.dilation) for rb in rbs}) == 1 and \
                        not windowed:
                    outs = self._resblocks_lockstep(rbs, [pk["rb"][i * nk + j] for j in range(nk)], a, axl)
                else:
                    outs = [self._resblock(rb, pk["rb"][i * nk + j], a, axl, pk["rbw"][i * nk + j]) for j, rb in enumerate(rbs)]
                if nk == 3:
                    al = ops.avg3(outs[0], outs[1], outs[2], 1.0 / 3.0, slope=nxt_slope)   # lrelu(xs / num_kernels)
                else:
                    raise NotImplementedError("MRF average is written for 3 resblock kernels per stage")
            wp, bp = pk["post"]
            self._mark("mrf%d" % (self.num_upsamples - 1))
            if al.shape[2] <= 128:
                y = ops.hifi_conv_post(al, wp, bp)                                         # conv_post -> tanh, streaming kernel
                self._mark("conv_post")
                return y
            Bn, Tout, C = al.shape
            y = torch.empty(Bn * Tout, 1, dtype=torch.float32, device=al.device)
            ops.conv1d(al, wp, bp, out=y.view(Bn, Tout, 1), flags=ops.TANH)
        return y.view(Bn, 1, Tout)
