"""FastSpeech2 on MI355X: the reference's module surface (`FastSpeech2(preprocess_config, model_config, n_speakers)`,
`forward(...) -> 12-tuple`, reference `state_dict` keys) over hand-written gfx950 kernels.

reference: fs_two/model/fastspeech2.py:12-119 (model graph), fs_two/transformer/Models.py (Encoder/Decoder),
Layers.py (FFTBlock, PostNet), SubLayers.py (MHA, FFN), model/modules.py (VarianceAdaptor, LengthRegulator,
VariancePredictor).  Every numbered step of `_forward` / `_backward` cites the lines it restates.

Forward AND backward are explicit sequences of kernel launches on the current HIP stream (no autograd graph
inside): activations are bf16 channels-last [B*T][C], accumulation is fp32, parameters are fp32 masters in one
flat buffer with a bf16 shadow.  `torch.autograd` only sees one node (`_Bridge`) so that the reference's
`loss.backward()` call site keeps working.
"""
import json
import os

import torch
import torch.nn as nn

from . import ops
from . import params as P
from . import switches
from .ops import bf16

N_VOCAB = 207  # len(symbols) + 1 — reference: fs_two/transformer/Models.py:40, fs_two/text/symbols.py


def sinusoid_table(n_position, d_hid):
    """reference: fs_two/transformer/Models.py:10-30 (float64, cast to fp32)."""
    pos = torch.arange(n_position, dtype=torch.float64)[:, None]
    j = torch.arange(d_hid)[None, :]
    angle = pos / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2.0 * (j // 2).double() / d_hid)
    tab = torch.empty_like(angle)
    tab[:, 0::2] = torch.sin(angle[:, 0::2])
    tab[:, 1::2] = torch.cos(angle[:, 1::2])
    return tab.float()


def get_speakers_number(preprocess_config):
    """reference: fs_two/model/fastspeech2.py:122-137."""
    path = os.path.join(preprocess_config["path"]["preprocessed_path"], "speakers.json")
    if not os.path.exists(path):
        raise Exception("Model is multispeaker but number of speakers was not provided explicitly")
    with open(path) as f:
        return len(json.load(f))


class _Bridge(torch.autograd.Function):
    """The single autograd node: forwards the natively computed outputs, routes their gradients into the native
    backward, which accumulates into the flat gradient buffer (the `.grad` views of the parameters)."""

    @staticmethod
    def forward(ctx, anchor, model, mel, post, pitch, energy, logd):
        ctx.model = model
        ctx.step_ctx = model._ctx
        return mel.view_as(mel), post.view_as(post), pitch.view_as(pitch), energy.view_as(energy), logd.view_as(logd)

    @staticmethod
    def backward(ctx, dmel, dpost, dpitch, denergy, dlogd):
        ctx.model._backward_from_autograd(ctx.step_ctx, dmel, dpost, dpitch, denergy, dlogd)
        return None, None, None, None, None, None, None


class _Ctx:
    """Activations kept for the backward of one step."""
    pass


class _GroupNotifier:
    """backward_native's contract with a data-parallel reducer: `done(name)` is called when every gradient of parameter group
    `name` has been ENQUEUED OR QUEUED; the queued part (grouped dW GEMMs, deferred reducers) must be flushed before the reducer
    may all-reduce the bucket.  Flushing at every group would cut the step's single grouped weight-gradient launch into 13; the
    notifier asks the reducer (`ready(name)`) whether a bucket actually completes with this group and only then flushes and
    announces.  Groups must arrive in `order` (checked: a reordering of backward would silently break the overlap contract)."""

    def __init__(self, order, on_bucket, flush, mark=None):
        self.order, self.on_bucket, self.flush = list(order), on_bucket, flush
        self.ready = getattr(getattr(on_bucket, "__self__", None), "ready", None)
        self.pos = 0
        self.flushes = 0
        # `mark(name)` (the "side" / "late" schedules): instead of flushing and announcing now, the caller records where the queues
        # stand and does both later, bucket by bucket (FastSpeech2._launch_dw_side_buckets / _flush_param_grads)
        self.mark = mark

    def done(self, name):
        if self.pos >= len(self.order) or self.order[self.pos] != name:
            raise RuntimeError("backward announced group %r, expected %r" % (name, self.order[self.pos] if self.pos < len(self.order) else None))
        self.pos += 1
        if self.on_bucket is None:
            return
        if self.mark is not None:
            self.mark(name)
            return
        if self.ready is not None and not self.ready(name):
            return                      # no gradient bucket completes with this group: keep queueing
        self.flush()
        self.flushes += 1
        self.on_bucket(name)


class FastSpeech2(nn.Module):
    def __init__(self, preprocess_config, model_config, n_speakers=None, device=None, seed=1234):
        super().__init__()
        self.model_config = model_config
        self.preprocess_config = preprocess_config
        if not model_config["multi_speaker"]:
            # reference: fastspeech2.py:72-88 raises NameError on this path; only multi-speaker works there too
            raise NotImplementedError("multi_speaker: False is not a working path in the reference either")
        if model_config["use_cwt"]:
            raise NotImplementedError("use_cwt: True (CWT pitch) is out of scope; shipped config sets False")
        if n_speakers is None:
            n_speakers = get_speakers_number(preprocess_config)
        if device is None or device == "gpu":
            device = "cuda:0" if torch.cuda.is_available() else "cpu"
        if isinstance(device, int):
            device = "cuda:%d" % device
        device = torch.device(device)
        tr = model_config["transformer"]
        self.d = tr["encoder_hidden"]
        self.d_ff = tr["conv_filter_size"]
        self.k1, self.k2 = tr["conv_kernel_size"]
        self.n_head_enc, self.n_head_dec = tr["encoder_head"], tr["decoder_head"]
        self.n_enc, self.n_dec = tr["encoder_layer"], tr["decoder_layer"]
        self.p_enc, self.p_dec = float(tr["encoder_dropout"]), float(tr["decoder_dropout"])
        self.p_var = float(model_config["variance_predictor"]["dropout"])
        self.k_var = model_config["variance_predictor"]["kernel_size"]
        self.p_post = 0.5                                                       # hard-coded in Layers.py:137-141
        self.max_seq_len = model_config["max_seq_len"]
        self.n_mel = preprocess_config["preprocessing"]["mel"]["n_mel_channels"]
        self.n_speakers = n_speakers
        assert self.d == 256 or self.d % 256 == 0, "LayerNorm kernel handles D in {256,512,768,1024}"

        entries = P.build_entries(model_config, self.n_mel, n_speakers, N_VOCAB)
        self._table, self._n_flat = P.layout(entries)
        self._flat = torch.zeros(self._n_flat, dtype=torch.float32, device=device)
        self._flat_grad = torch.zeros(self._n_flat, dtype=torch.float32, device=device)
        self._shadow = torch.zeros(self._n_flat, dtype=bf16, device=device)
        self._shadow_version = -1
        self._anchor = torch.zeros((), requires_grad=True)
        self._ctx = None
        self._deferred = None           # split-K slabs awaiting the batched reducer (backward only)
        self._deferred_fin = None       # gradient column-sum partials awaiting the batched finalize
        self._side = None               # second HIP stream for parameter-gradient work (see _SideWork)
        # The FFT blocks' w_1 forward, q|k|v projection, fc and w_2 input gradients, and the PostNet's 512 -> 512 convs (forward and
        # input gradient) run on the window kernel (csrc/ffn_conv.hip).  It wants the weights in MFMA-fragment order (1 KiB contiguous
        # per fragment; from the tap-major shadow it is no faster than the implicit GEMM): `_w1_packed` holds such copies — for an input
        # gradient the transposed, tap-flipped weight — all rewritten by ONE launch whenever the bf16 shadow is (sync_shadow, the
        # optimizer step).
        self.window_ffn = switches.get("TTSK_WINDOW_FFN") != "0"
        self._w1_packed = None
        self._adam_tables = None
        self.adam_packs = True          # the optimizer's Adam launch writes the window kernels' weight packs itself (ttsk_optim_step_packed)
        self.flash_attention = True     # attention without the S x S tensors when d_k = 128 (csrc/flash_attn.hip); False / other head sizes: scores GEMM + softmax + P V GEMM
        self.fused_ln = True            # fc / w_2 + dropout + residual + LayerNorm + PAD zeroing in one kernel when d = 256
        self._postnet_ends_win = True       # the PostNet's 80 -> 512 / 512 -> 80 convs and their input gradients on the window kernel too (round 5; property below)
        self.bn_stats_in_conv = True        # PostNet 512 -> 512 convs emit their BatchNorm statistics partials
        self.bn_bwd_stats_in_conv = True    # ... and their input-gradient convs the backward's
        self.fused_qkv_tail = True          # a block's last kernel also projects q|k|v for the next block
        self.fused_qkv_dx = True            # ... and the q|k|v input gradient of the block behind in front of it
        self.fused_ln_bwd = True            # LayerNorm backward + the k = 1 dX projection behind it in one kernel
        self.dwconv = switches.get("TTSK_DWCONV") != "0"   # w_1's weight gradient on the tap-sharing kernel (csrc/dwconv.hip), the other 256-multiple ones on dwgemm.hip
        self.group_predictors = True    # training with targets: the three VariancePredictors run as grouped launches
        self.raw_slabs = True           # dX GEMMs that feed a LayerNorm backward leave their split-K tiles for it to sum
        self.group_param_grads = True      # weight-gradient GEMMs of a backward pass share grouped launches (ops.DeferQueue)
        self.overlap_param_grads = False   # measured on MI355X: the branches do overlap under graph replay, but the concurrent
                                           # kernels slow each other by as much (6.39 vs 6.47 ms/step): off by default
        # The weight-gradient GEMMs of PostNet + decoder (88 % of the step's dW FLOPs) are complete once the decoder's backward is;
        # what follows on the dX path (length regulator, variance adaptor, encoder: ~0.5 ms of 32-128-workgroup kernels) fills a
        # fraction of the chip.  So that group is launched there on a second stream with its grid capped at `dw_side_wgs` workgroups
        # (one per CU: the other CUs stay free for the dX chain) and joined before the optimizer.  0 = launch it at the end instead.
        self.dw_side_wgs = int(switches.get("TTSK_DW_SIDE_WGS"))
        # ... and only `dw_side_frac` of that group's FLOPs go there: the dX chain that runs beside it is shorter (0.39 ms) than the
        # capped group (0.57 ms), the rest joins the encoder-side group that runs on the whole chip once the dX chain is done
        # (measured: 1.0 -> 3.03 ms/step, 0.85 / 0.75 -> 3.00, 0.65 -> 3.02, 0.55 -> 3.11)
        # (round 3: w_1's gradients left the grouped queue for csrc/dwconv.hip, which runs first on that stream: re-measured below)
        self.dw_side_frac = float(switches.get("TTSK_DW_SIDE_FRAC"))
        self._dw_side = None
        self._dw_side_pending = False
        # Data-parallel schedule (backward_native(on_bucket=...)), TTSK_DP_SCHEDULE:
        #   "side"  (default) the single-GPU schedule kept: nothing is flushed during the PostNet / decoder backward; after it the queued
        #           weight-gradient work runs on the second stream (capped grid, its split-K reducer behind it) and the all-reduces of
        #           the buckets it completes are issued from there, beside the encoder-side dX chain; the rest after the chain;
        #   "late"  the single-GPU schedule untouched, every all-reduce after the last flush (no overlap with backward).
        # (Round 2's "early" schedule — a flush on the main stream whenever a bucket completes — measured 22 % more compute per step and
        # was deleted in round 4; without a second stream, dw_side_wgs = 0, buckets are still flushed one by one as they complete.)
        self.dp_schedule = switches.get("TTSK_DP_SCHEDULE")
        self.side_small = False             # the 80-channel grouped problems behind dwgemm on the second stream (measured: they belong in the final phase)
        self._fin_side = None
        self._dp_keep = None
        self._fin_pending = False
        # Training with targets: the three VariancePredictors' outputs feed nothing but the loss (the embeddings are picked by the TARGET
        # pitch / energy, the length regulator takes the TARGET durations: modules.py:158-205), and their backward needs nothing but the
        # loss's gradients until its last step.  Both run on a stream of their own: the forward beside the decoder's first block, the
        # backward beside the PostNet's — 1,024-row kernels of 96 workgroups that the 212-256-workgroup chain kernels leave room for.
        self.pred_side = switches.get("TTSK_PRED_SIDE")     # "1" both, "f" forward only, "b" backward only, "0" neither
        self._pred_stream = None
        self._pred_fwd_pending = False
        self._var_on_pred = False           # the loss was taken in two halves (graph.make_enqueue): the variance gradients live on the predictors' stream
        self._loss_finalize = None          # ... and this makes the loss values, on a stream that has waited for both halves (backward_native)
        self.split_loss = True              # training steps built by graph.make_enqueue take the loss that way when the predictors have their stream
        # Does the flat gradient buffer hold an unfinished accumulation (micro-steps of a grad_acc_step cycle)?  False after an optimizer
        # update or zero_grad(): the next backward then overwrites instead of accumulating (see backward_native).
        self.grads_partial = False
        self._acc = True
        self._rng_state = None          # device block shared with the optimizer (ops.optim_state)
        self._seed = seed
        self._modules_by_key = {}
        for en in entries:
            self._register(en, device)
        self._init_constants(preprocess_config, model_config)
        self.reset_parameters(seed)
        # the three predictors sit back to back in the flat buffer with identical internal layouts: constant stride
        va = "variance_adaptor."
        o = [self._table[va + n + "_predictor.conv_layer.conv1d_1.conv.weight"].offset for n in ("duration", "pitch", "energy")]
        self._pred_stride = o[1] - o[0]
        assert o[2] - o[1] == self._pred_stride and self._pred_stride % 8 == 0
        if self.window_ffn and self._shadow.is_cuda:
            self._build_packs()

    # ------------------------------------------------------------------ parameter plumbing
    def _register(self, en, device):
        mod = self
        parts = en.key.split(".")
        for name in parts[:-1]:
            if name not in mod._modules:
                mod.add_module(name, nn.Module())
            mod = mod._modules[name]
        leaf = parts[-1]
        if en.kind == P.TRAIN:
            par = nn.Parameter(self._view(self._flat, en), requires_grad=True)
            par.grad = self._view(self._flat_grad, en)
            mod.register_parameter(leaf, par)
        elif en.kind in (P.FROZEN, P.UNUSED):
            mod.register_parameter(leaf, nn.Parameter(torch.zeros(en.shape, device=device), requires_grad=en.kind == P.UNUSED))
        else:
            dt = torch.int64 if en.dtype == "int64" else torch.float32
            mod.register_buffer(leaf, torch.zeros(en.shape, dtype=dt, device=device))
        self._modules_by_key[en.key] = (mod, leaf)

    @staticmethod
    def _view(flat, en):
        v = flat[en.offset:en.offset + en.numel].view(en.storage_shape)
        return v.permute(0, 2, 1) if en.conv else v

    def _rebind(self):
        """After the flat buffers moved (`.to()`, `.cuda()`): new Parameters over the new views.  `par.data = view` would keep
        the parameter's OWN version counter, so a later `load_state_dict` / `param.copy_()` would no longer bump
        `_flat._version` and the bf16 shadow would go stale; a Parameter built from a view shares the buffer's counter."""
        for en in self._table.values():
            if en.kind == P.TRAIN:
                mod, leaf = self._modules_by_key[en.key]
                par = nn.Parameter(self._view(self._flat, en), requires_grad=True)
                par.grad = self._view(self._flat_grad, en)
                mod._parameters[leaf] = par
        self._shadow_version = -1

    def mark_dirty(self):
        """Call after writing the fp32 masters through anything torch cannot see (raw pointers): the next forward
        refreshes the bf16 shadow."""
        self._shadow_version = -1

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._shadow_version = -1
        return out

    def _apply(self, fn, recurse=True):
        flat, grad, shadow = fn(self._flat), fn(self._flat_grad), fn(self._shadow)
        if flat.dtype != torch.float32:
            raise TypeError("FastSpeech2 masters stay fp32 (bf16 shadows are managed internally)")
        super()._apply(fn, recurse)
        self._flat, self._flat_grad = flat, grad
        self._shadow = shadow.to(bf16) if shadow.dtype != bf16 else shadow
        self._w1_packed = None
        self._adam_tables = None
        self._shadow_version = -1
        self._rng_state = None
        self._side = self._dw_side = self._fin_side = self._pred_stream = None
        self._rebind()
        if self.window_ffn and self._shadow.is_cuda:
            self._build_packs()
        return self

    def get(self, key):
        mod, leaf = self._modules_by_key[key]
        return mod._parameters[leaf] if leaf in mod._parameters else mod._buffers[leaf]

    def _init_constants(self, pc, mc):
        tab = sinusoid_table(mc["max_seq_len"] + 1, self.d)[None]
        with open(os.path.join(pc["path"]["preprocessed_path"], "stats.json")) as f:     # modules.py:55-60
            stats = json.load(f)
        nb = mc["variance_embedding"]["n_bins"]
        ve = mc["variance_embedding"]
        with torch.no_grad():
            self.get("encoder.position_enc").copy_(tab)
            self.get("decoder.position_enc").copy_(tab)
            for name, key in (("pitch", "pitch_bins"), ("energy", "energy_bins")):
                lo, hi = stats[name][:2]
                if ve[name + "_quantization"] == "log":
                    import numpy as np
                    bins = torch.exp(torch.linspace(np.log(lo), np.log(hi), nb - 1))
                else:
                    bins = torch.linspace(lo, hi, nb - 1)
                self.get("variance_adaptor." + key).copy_(bins)
            for i in range(5):
                self.get("postnet.convolutions.%d.1.running_var" % i).fill_(1.0)

    def reset_parameters(self, seed=1234):
        """Seeded random init (weights_path: null).  Statistics follow tts_king_amd.synthetic.seeded_fill;
        the PAD row of the phoneme embedding is zero as in nn.Embedding(padding_idx=0)."""
        from .synthetic import seeded_fill
        sd = {k: v for k, v in self.state_dict().items()}
        cpu = {k: torch.zeros(v.shape, dtype=v.dtype) for k, v in sd.items()}
        for k, v in sd.items():
            if any(s in k for s in ("position_enc", "_bins", "running_", "num_batches")):
                cpu[k] = v.detach().cpu().clone()
        seeded_fill(cpu, seed)
        cpu["encoder.src_word_emb.weight"][0].zero_()
        with torch.no_grad():
            for k, v in sd.items():
                if not any(s in k for s in ("position_enc", "_bins", "running_", "num_batches")):
                    v.copy_(cpu[k])

    @property
    def device(self):
        return self._flat.device

    def flat_buffers(self):
        """(params fp32, grads fp32, bf16 shadow): the trainable state as three flat tensors."""
        return self._flat, self._flat_grad, self._shadow

    def grad_buckets(self, bucket_mb=24):
        return P.buckets(self._table, self._n_flat, int(bucket_mb * (1 << 20) / 4))

    def group_offsets(self):
        """name -> lowest flat offset of the parameter group announced by backward_native(on_bucket=...)."""
        def lo(prefixes):
            return min(en.offset for k, en in self._table.items() if en.kind == P.TRAIN and k.startswith(prefixes))
        g = {"postnet": lo(("postnet.",)), "mel_linear": lo(("mel_linear.",)),
             "variance_adaptor": lo(("variance_adaptor.", "speaker_emb.")), "embedding": 0}
        for i in range(self.n_dec):
            g["decoder.%d" % i] = lo(("decoder.layer_stack.%d." % i,))
        for i in range(self.n_enc):
            g["encoder.%d" % i] = lo(("encoder.layer_stack.%d." % i,))
        return g

    def attach_state(self, state):
        """Share the optimizer's device state block (dropout counters live in it)."""
        self._rng_state = state

    def _state(self):
        if self._rng_state is None:
            self._rng_state = ops.optim_state(self.device, seed=self._seed)
        return self._rng_state

    def sync_shadow(self, force=False):
        """bf16 copies of the masters, refreshed when anything wrote to them through torch (load_state_dict, init)."""
        if force or self._shadow_version != self._flat._version:
            ops.cast_bf16(self._flat, self._shadow)
            self._shadow_version = self._flat._version
            self.refresh_packed()

    @property
    def postnet_ends_win(self):
        return self._postnet_ends_win

    @postnet_ends_win.setter
    def postnet_ends_win(self, on):
        """The attribute decides which packs exist, so a change after construction rebuilds them (it was only read in `_build_packs`)."""
        on = bool(on)
        if on != self._postnet_ends_win:
            self._postnet_ends_win = on
            self._w1_packed = None
            self._adam_tables = None
            if self.window_ffn and self._shadow.is_cuda:
                self._build_packs()

    def refresh_packed(self):
        """Rewrite the fragment-major weight copies the window conv kernel reads (csrc/ffn_conv.hip) from the bf16 shadow — one launch
        over a device-resident item table.  Called by everything that writes the shadow: `sync_shadow` and
        `ScheduledOptim.step_and_update_lr`."""
        if not self.window_ffn:
            return
        if self._w1_packed is None:
            self._build_packs()
        if self._pack_table is not None:
            ops.win_conv_pack_run(*self._pack_table)
        self.refresh_odd_packs()

    def refresh_odd_packs(self):
        """The packs the optimizer's Adam launch does not write itself (the PostNet's 80-channel ends), from the bf16 shadow: one small launch."""
        if self.window_ffn and getattr(self, "_odd_pack_table", None) is not None:
            ops.win_conv_pack_run(*self._odd_pack_table)

    def _build_packs(self):
        """Allocate the packs and their item table (host -> device copy: not capturable, so this runs at construction and after
        `_apply`, never inside a forward).  Packs: the FFT blocks' w_1 and q|k|v as they are (forward), fc / w_2 / w_1 / q|k|v transposed
        with flipped taps (input gradients), the PostNet's three 512 -> 512 convs both ways."""
        d, dec = self.d, ["decoder.layer_stack.%d." % i for i in range(self.n_dec)]
        enc = ["encoder.layer_stack.%d." % i for i in range(self.n_enc)]
        # (tag, key, rows of a fused view or None, transpose)
        want = [("w1", p + "pos_ffn.w_1.weight", None, False) for p in dec + enc] + \
               [("qkv", p + "slf_attn.w_qs.weight", 3 * d, False) for p in dec + enc] + \
               [("fcT", p + "slf_attn.fc.weight", None, True) for p in dec + enc] + \
               [("w2T", p + "pos_ffn.w_2.weight", None, True) for p in dec + enc] + \
               [("w1T", p + "pos_ffn.w_1.weight", None, True) for p in dec + enc] + \
               [("fc", p + "slf_attn.fc.weight", None, False) for p in dec + enc] + \
               [("w2", p + "pos_ffn.w_2.weight", None, False) for p in dec + enc] + \
               [("qkvT", p + "slf_attn.w_qs.weight", 3 * d, True) for p in dec + enc] + \
               [("pn", "postnet.convolutions.%d.0.conv.weight" % i, None, False) for i in range(1, 4)] + \
               [("pnT", "postnet.convolutions.%d.0.conv.weight" % i, None, True) for i in range(1, 4)]
        # the PostNet's ends (80 -> 512 and 512 -> 80 channels, round 5): packs Adam's tile walk cannot write (80 is neither a multiple of
        # 32 storage rows nor of 256 storage columns); one small launch of their own behind the optimizer step (refresh_odd_packs)
        odd = [(t, "postnet.convolutions.%d.0.conv.weight" % i, None, tr) for t, i, tr in
               (("pn", 0, False), ("pnT", 0, True), ("pn", 4, False), ("pnT", 4, True))] if self.postnet_ends_win else []
        if self.postnet_ends_win:            # ... and mel_linear (256 -> 80) both ways
            odd += [("lin", "mel_linear.weight", None, False), ("linT", "mel_linear.weight", None, True)]
        items = []
        for tag, key, fused_rows, tr in want:
            if key not in self._table:
                continue
            W = self._pack_source(key, fused_rows)
            cs, kk, ds = W.shape
            cin, cout = (cs, ds) if tr else (ds, cs)
            split = tag in ("w1T", "qkvT")                 # wide contraction: 256-channel slices into fp32 slabs (ops.win_conv_split)
            if tag in ("fc", "w2"):                        # the fused projection + LayerNorm kernel's weights (ops.win_ln_fwd)
                if kk != 1 or not ops.win_ln_supported(cin, cout):
                    continue
            elif not ops.win_conv_supported(256 if split else cin, cout, kk) or (split and cin % 256):
                continue
            items.append((tag, key, fused_rows, tr, W.numel()))
        buf = torch.empty(sum(it[4] for it in items), dtype=bf16, device=self._shadow.device)
        self._w1_packed, self._pack_items, off = {}, [], 0
        for tag, key, fused_rows, tr, n in items:
            self._w1_packed[(tag, key)] = buf[off:off + n]
            self._pack_items.append((key, fused_rows, buf[off:off + n], tr))
            off += n
        # the shadow views and the packs keep their addresses until _apply: a device-resident item table, one pack launch per step
        self._pack_table = ops.win_conv_pack_table([(self._pack_source(key, fr), out, tr) for key, fr, out, tr in self._pack_items],
                                                   self._shadow.device) if self._pack_items else None
        self._odd_pack_table = None
        odd_items = []
        for tag, key, _, tr in odd:
            if key not in self._table:
                continue
            W = self._pack_source(key)
            cs, kk, ds = W.shape
            cin, cout = (cs, ds) if tr else (ds, cs)
            if not ops.win_conv_supported(cin, cout, kk):
                continue
            pk = torch.empty(ops.win_pack_numel(cs, kk, ds, tr), dtype=bf16, device=self._shadow.device)
            self._w1_packed[(tag, key)] = pk
            odd_items.append((W, pk, tr))
        if odd_items:
            self._odd_pack_table = ops.win_conv_pack_table(odd_items, self._shadow.device)
        # ... and the tables with which the optimizer's Adam launch writes these packs itself (ttsk_optim_step_packed): per weight its
        # flat offset, storage shape and its plain / transposed packs.  A weight with more than one pack of a kind (w_1's transposed
        # pack exists once) or one that does not tile leaves `_adam_tables` None: the optimizer then calls refresh_packed as before.
        self._adam_tables = None
        if self._pack_items and self.adam_packs:
            by_key, ok = {}, True
            for key, fr, out, tr in self._pack_items:
                W = self._pack_source(key, fr)
                ent = by_key.setdefault(key, [self._table[key].offset, tuple(W.shape), None, None])
                if ent[1] != tuple(W.shape) or ent[3 if tr else 2] is not None:
                    ok = False
                ent[3 if tr else 2] = out
            if ok:
                self._adam_tables = ops.adam_pack_tables([tuple(v) for v in by_key.values()], self._n_flat, self._shadow.device)

    def _pack_source(self, key, fused_rows=None):
        """The tap-major bf16 shadow of `key` as a (Cs, k, Ds) tensor (a Linear weight is k = 1; `fused_rows`: the q|k|v rows as one)."""
        W = self._w(key, fused_rows) if fused_rows is not None else self._w(key)
        return W.view(W.shape[0], 1, W.shape[1]) if W.dim() == 2 else W

    # views into the flat buffers ------------------------------------------------------------------
    def _w(self, key, rows=None):
        """bf16 shadow of a weight as a 2-D/3-D tensor in STORAGE layout; `rows` widens it over the following
        keys (fused q|k|v)."""
        en = self._table[key]
        shp = en.storage_shape
        if rows is not None:
            return self._shadow[en.offset:en.offset + rows * shp[1]].view(rows, shp[1])
        return self._shadow[en.offset:en.offset + en.numel].view(shp)

    def _m(self, key, n=None):
        """fp32 master (vectors / tables) — `n` widens over the following keys."""
        en = self._table[key]
        if n is not None:
            return self._flat[en.offset:en.offset + n]
        return self._flat[en.offset:en.offset + en.numel].view(en.storage_shape)

    def _g(self, key, n=None):
        en = self._table[key]
        if n is not None:
            return self._flat_grad[en.offset:en.offset + n]
        return self._flat_grad[en.offset:en.offset + en.numel].view(en.storage_shape)

    # ------------------------------------------------------------------ forward
    def forward(self, speakers, texts, src_lens, max_src_len, mels=None, mel_lens=None, max_mel_len=None,
                e_targets=None, d_targets=None, pitches_raw=None, pitches_cwt=None, pitches_mean=None,
                pitches_std=None, p_control=1.0, e_control=1.0, d_control=1.0):
        """reference: fs_two/model/fastspeech2.py:43-119.  Same arguments, same 12-tuple."""
        if not self._flat.is_cuda:
            raise ops.L.TtskError("FastSpeech2.forward needs the model on a HIP device (gpu: 'cuda:0'); there is no CPU path")
        train = self.training
        dev = self.device
        speakers, texts = speakers.to(dev).long().contiguous(), texts.to(dev).long().contiguous()
        src_lens = src_lens.to(dev).long().contiguous()
        with torch.no_grad():
            out, ctx = self._forward(train, speakers, texts, src_lens, int(max_src_len), mel_lens, max_mel_len, e_targets,
                                     d_targets, pitches_raw, float(p_control), float(e_control), float(d_control))
        mel, pitch, energy, logd, d_rounded, src_masks, mel_masks, mel_lens_out, post = out
        self._ctx = ctx
        if train and torch.is_grad_enabled():
            mel, post, pitch, energy, logd = _Bridge.apply(self._anchor, self, mel, post, pitch, energy, logd)
        return (mel, pitch, energy, logd, d_rounded, src_masks, mel_masks, src_lens, mel_lens_out, post, None, None)

    def _fft_fwd(self, pre, x, Bn, S, lens, H, p, site, rng, ctx_list, out=None, qkv=None, next_pre=None):
        """One FFTBlock.  reference: Layers.py:25-34, SubLayers.py:31-65 (MHA), :93-101 (FFN), Modules.py:14-24.
        `next_pre`: the block that follows in the same stack — its q|k|v projection of this block's output comes out of this
        block's last kernel (then the return value is (x2, qkv_next), and the caller hands qkv_next to that block as `qkv`)."""
        d, rows = self.d, Bn * S
        dk = d // H
        a, f = pre + "slf_attn.", pre + "pos_ffn."
        Sp = (S + 7) // 8 * 8
        dev = x.device
        # (1) q|k|v projections as one GEMM into a [rows][3d] buffer: SubLayers.py:41-43
        pkq = self._w1_packed.get(("qkv", a + "w_qs.weight")) if (self.window_ffn and self._w1_packed) else None
        if qkv is not None:
            pass                         # came out of the previous block's last kernel
        elif pkq is not None and x.dtype == bf16:
            qkv = ops.win_conv(x.view(Bn, S, d), pkq, 3 * d, 1, bias=self._m(a + "w_qs.bias", 3 * d)).view(rows, 3 * d)   # window kernel, k = 1
        else:
            qkv = ops.linear(x, self._w(a + "w_qs.weight", 3 * d), self._m(a + "w_qs.bias", 3 * d))
        if self.flash_attention and dk == 128:
            # (2)-(4) one kernel, no S x S tensor: scores, key-padding mask, online softmax, P V, heads merged (Modules.py:15-22,
            # SubLayers.py:57-60); the backward recomputes P from the per-row log-sum-exp kept in `probs`'s slot
            o, probs, o32 = ops.flash_attention_fwd(qkv, lens, Bn, H, S, want_lse=ctx_list is not None)
        else:
            o32 = None
            # (2) scores = Q K^T / sqrt(dk) per (batch, head), head h = columns [h*dk, (h+1)*dk): Modules.py:15-16
            scores = torch.empty(Bn * H, S, Sp, dtype=torch.float32, device=dev)
            ops.gemm(qkv, qkv[:, d:], scores, S, S, dk, 3 * d, 3 * d, Sp, alpha=dk ** -0.5, nz1=Bn, nz2=H,
                     sA=(S * 3 * d, dk), sB=(S * 3 * d, dk), sC=(H * S * Sp, S * Sp))
            # (3) key-padding mask + softmax: Modules.py:18-21
            probs = ops.softmax_fwd(scores, lens, H)
            # (4) O = P V, heads merged back into [rows][d]: Modules.py:22, SubLayers.py:57-60
            o = torch.empty(rows, d, dtype=bf16, device=dev)
            ops.gemm(probs, qkv[:, 2 * d:], o, S, dk, S, Sp, 3 * d, d, flags=ops.B_TR, nz1=Bn, nz2=H,
                     sA=(H * S * Sp, S * Sp), sB=(S * 3 * d, dk), sC=(S * d, dk))
        fuse = self.fused_ln and d == 256
        qkv_next = None
        # (5) fc, dropout, +residual, LayerNorm, zero PAD rows: SubLayers.py:62-63, Layers.py:29 — one kernel when d = 256
        if fuse:
            pkl = self._w1_packed.get(("fc", a + "fc.weight")) if (self.window_ffn and self._w1_packed) else None
            if pkl is not None:
                x1, z1, mean1, rstd1 = ops.win_ln_fwd(o, pkl, self._m(a + "fc.bias"), x, self._m(a + "layer_norm.weight"),
                                                      self._m(a + "layer_norm.bias"), lens, S, p_pre=p, site_pre=site, rng=rng,
                                                      save_z=ctx_list is not None)
            else:
                x1, z1, mean1, rstd1 = ops.gemm_ln_fwd(o, self._w(a + "fc.weight"), self._m(a + "fc.bias"), x, self._m(a + "layer_norm.weight"),
                                                       self._m(a + "layer_norm.bias"), lens, S, p_pre=p, site_pre=site, rng=rng,
                                                       save_z=ctx_list is not None)
        else:
            y = ops.linear(o, self._w(a + "fc.weight"), self._m(a + "fc.bias"))
            x1, z1, mean1, rstd1, _ = ops.layernorm_fwd(y, x, self._m(a + "layer_norm.weight"), self._m(a + "layer_norm.bias"),
                                                        lens, S, p_pre=p, site_pre=site, rng=rng, save_z=ctx_list is not None)
        # (6) FFN: Conv1d(k=9)+ReLU, Conv1d(k=1), dropout, +residual, LayerNorm, zero PAD rows: SubLayers.py:96-99, Layers.py:32
        W1 = self._w(f + "w_1.weight")
        pk = self._w1_packed.get(("w1", f + "w_1.weight")) if (self.window_ffn and self._w1_packed) else None
        if pk is not None and x1.dtype == bf16:
            h = ops.ffn_conv_fwd(x1.view(Bn, S, d), W1, self._m(f + "w_1.bias"), relu=True, packed=pk)      # window kernel (csrc/ffn_conv.hip)
        else:
            h = ops.conv1d(x1.view(Bn, S, d), W1, self._m(f + "w_1.bias"), flags=ops.RELU)
        if fuse and self.k2 == 1:
            pkl = self._w1_packed.get(("w2", f + "w_2.weight")) if (self.window_ffn and self._w1_packed) else None
            pkn = self._w1_packed.get(("qkv", next_pre + "slf_attn.w_qs.weight")) if (next_pre and pkl is not None and self.fused_qkv_tail) else None
            if pkn is not None:
                x2, z2, mean2, rstd2, qkv_next = ops.win_ln_fwd(h.view(rows, -1), pkl, self._m(f + "w_2.bias"), x1, self._m(f + "layer_norm.weight"),
                                                                self._m(f + "layer_norm.bias"), lens, S, p_pre=p, site_pre=site + 1, rng=rng,
                                                                save_z=ctx_list is not None, out=out,
                                                                proj=(pkn, self._m(next_pre + "slf_attn.w_qs.bias", 3 * d)))
            elif pkl is not None:
                x2, z2, mean2, rstd2 = ops.win_ln_fwd(h.view(rows, -1), pkl, self._m(f + "w_2.bias"), x1, self._m(f + "layer_norm.weight"),
                                                      self._m(f + "layer_norm.bias"), lens, S, p_pre=p, site_pre=site + 1, rng=rng,
                                                      save_z=ctx_list is not None, out=out)
            else:
                x2, z2, mean2, rstd2 = ops.gemm_ln_fwd(h.view(rows, -1), self._w(f + "w_2.weight"), self._m(f + "w_2.bias"), x1,
                                                       self._m(f + "layer_norm.weight"), self._m(f + "layer_norm.bias"), lens, S, p_pre=p,
                                                       site_pre=site + 1, rng=rng, save_z=ctx_list is not None, out=out)
        else:
            y2 = ops.conv1d(h, self._w(f + "w_2.weight"), self._m(f + "w_2.bias"))
            x2, z2, mean2, rstd2, _ = ops.layernorm_fwd(y2.view(rows, d), x1, self._m(f + "layer_norm.weight"),
                                                        self._m(f + "layer_norm.bias"), lens, S, p_pre=p, site_pre=site + 1,
                                                        rng=rng, save_z=ctx_list is not None, out=out)
        if ctx_list is not None:
            ctx_list.append((pre, x, qkv, probs, o, z1, mean1, rstd1, x1, h, z2, mean2, rstd2, Bn, S, lens, H, p, site, o32))
        if next_pre:
            return x2, qkv_next
        return x2

    def _predictor_fwd(self, pre, x, Bn, Lp, lens, p, site, rng, ctx):
        """VariancePredictor.  reference: model/modules.py:255-309."""
        d, rows = self.d, Bn * Lp
        c = pre + "conv_layer."
        h1 = ops.conv1d(x.view(Bn, Lp, d), self._w(c + "conv1d_1.conv.weight"), self._m(c + "conv1d_1.conv.bias"), flags=ops.RELU)
        a1, _, m1, r1, _ = ops.layernorm_fwd(h1.view(rows, -1), None, self._m(c + "layer_norm_1.weight"), self._m(c + "layer_norm_1.bias"),
                                             None, 0, p_post=p, site_post=site, rng=rng, save_z=False)
        h2 = ops.conv1d(a1.view(Bn, Lp, -1), self._w(c + "conv1d_2.conv.weight"), self._m(c + "conv1d_2.conv.bias"), flags=ops.RELU)
        _, _, m2, r2, out = ops.layernorm_fwd(h2.view(rows, -1), None, self._m(c + "layer_norm_2.weight"), self._m(c + "layer_norm_2.bias"),
                                              lens, Lp, p_post=p, site_post=site + 1, rng=rng, save_z=False, want_out=False,
                                              head=(self._m(pre + "linear_layer.weight").view(-1), self._m(pre + "linear_layer.bias")))
        if ctx is not None:
            ctx[pre] = (x, h1, m1, r1, a1, h2, m2, r2, Bn, Lp, lens, p, site)
        return out.view(Bn, Lp)

    def _predictors_fwd_grouped(self, stack, Bn, Lp, lens, p, rng, ctx, row_limit=None):
        """The duration / pitch / energy VariancePredictors (model/modules.py:255-309) as ONE chain of grouped launches: with
        targets given (training) their inputs x, x + speaker, x + speaker + pitch_emb[target] do not depend on each other's
        outputs (modules.py:158-193).  stack (3, rows, d) bf16 holds the three inputs.  Returns (3, B, L) fp32 = (log-duration,
        pitch, energy) predictions."""
        d, rows, ps = self.d, Bn * Lp, self._pred_stride
        pre = "variance_adaptor.duration_predictor."
        c = pre + "conv_layer."
        W1, W2 = self._w(c + "conv1d_1.conv.weight"), self._w(c + "conv1d_2.conv.weight")
        Fh = W1.shape[0]
        dev = stack.device
        h1 = torch.empty(3, rows, Fh, dtype=bf16, device=dev)
        ops.conv1d(stack[0].view(Bn, Lp, d), W1, self._m(c + "conv1d_1.conv.bias"), flags=ops.RELU, out=h1[0].view(Bn, Lp, Fh),
                   nz1=3, sA=(rows * d, 0), sB=(ps, 0), sC=(rows * Fh, 0), s_bias1=ps)
        # row_limit (bucketed L): hidden rows past the batch's own longest text are zero rows — the second conv's zero padding
        a1, m1, r1, _ = ops.layernorm_fwd_grouped(h1.view(3 * rows, Fh), self._m(c + "layer_norm_1.weight"), self._m(c + "layer_norm_1.bias"),
                                                  3, ps, 2, row_limit, Lp if row_limit is not None else 0, p_post=p, site_post=200, rng=rng)
        h2 = torch.empty(3, rows, Fh, dtype=bf16, device=dev)
        ops.conv1d(a1[:rows].view(Bn, Lp, Fh), W2, self._m(c + "conv1d_2.conv.bias"), flags=ops.RELU, out=h2[0].view(Bn, Lp, Fh),
                   nz1=3, sA=(rows * Fh, 0), sB=(ps, 0), sC=(rows * Fh, 0), s_bias1=ps)
        _, m2, r2, out = ops.layernorm_fwd_grouped(h2.view(3 * rows, Fh), self._m(c + "layer_norm_2.weight"), self._m(c + "layer_norm_2.bias"),
                                                   3, ps, 2, lens, Lp, p_post=p, site_post=201, rng=rng, want_out=False,
                                                   head=(self._m(pre + "linear_layer.weight").view(-1), self._m(pre + "linear_layer.bias")))
        if ctx is not None:
            ctx["grouped"] = (stack, h1, m1, r1, a1, h2, m2, r2, Bn, Lp, lens, p, row_limit)
        return out.view(3, Bn, Lp)

    def _predictors_bwd_grouped(self, saved, dstack, rng, dx3, dxin=None):
        """Backward of _predictors_fwd_grouped; dstack (3, B, L) fp32 = gradients of (log-duration, pitch, energy) predictions,
        dx3 = gradient of the LengthRegulator input.  Returns (dx2, dx1, dx): gradients of x2 (what pitch_embedding collects),
        x1 (speaker_emb) and of the encoder output.  `dxin`: the three predictors' input gradients if _predictors_bwd_inputs ran already."""
        if dxin is None:
            dxin = self._predictors_bwd_inputs(saved, dstack, rng)
        Lp, row_limit = saved[9], saved[12]
        return ops.va_combine(dx3, dxin, Lp, row_limit)

    def _predictors_bwd_inputs(self, saved, dstack, rng):
        """Everything of the predictors' backward that needs only the loss's gradients: (3, rows, d) fp32 input gradients, and the
        parameter-gradient work queued."""
        (stack, h1, m1, r1, a1, h2, m2, r2, Bn, Lp, lens, p, row_limit) = saved
        d, rows, ps = self.d, Bn * Lp, self._pred_stride
        names = ("duration", "pitch", "energy")
        pre = "variance_adaptor.duration_predictor."
        c = pre + "conv_layer."
        W1, W2 = self._w(c + "conv1d_1.conv.weight"), self._w(c + "conv1d_2.conv.weight")
        Fh = W1.shape[0]
        dev = dstack.device
        dh2, part, nblk = ops.layernorm_bwd_grouped(None, h2.view(3 * rows, Fh), m2, r2, self._m(c + "layer_norm_2.weight"),
                                                    self._m(c + "layer_norm_2.bias"), 3, ps, 2, lens, Lp, relu_in=True, p_post=p,
                                                    site_post=201, rng=rng, dhead=dstack.view(-1),
                                                    head_w=self._m(pre + "linear_layer.weight").view(-1))
        dh2 = dh2.view(3, rows, Fh)
        with self._side_work(dh2, part, a1):
            for g, n in enumerate(names):
                cg = "variance_adaptor.%s_predictor.conv_layer." % n
                self._finalize_ln(part[g], nblk, 4 * Fh + 1, cg + "conv1d_2.conv.bias")
                ops.queue_dw(self._deferred, dh2[g].view(Bn, Lp, Fh), a1[g * rows:(g + 1) * rows].view(Bn, Lp, Fh), self._g(cg + "conv1d_2.conv.weight"),
                             None, self._acc, k=self.k_var, use_dwgemm=self._use_dwconv)
        da1 = torch.empty(3, rows, Fh, dtype=bf16, device=dev)
        ops.conv1d_dx(dh2[0].view(Bn, Lp, Fh), W2, out=da1[0].view(Bn, Lp, Fh), nz1=3, sA=(rows * Fh, 0), sB=(ps, 0), sC=(rows * Fh, 0))
        dh1, part, nblk = ops.layernorm_bwd_grouped(da1.view(3 * rows, Fh), h1.view(3 * rows, Fh), m1, r1, self._m(c + "layer_norm_1.weight"),
                                                    self._m(c + "layer_norm_1.bias"), 3, ps, 2, row_limit, Lp if row_limit is not None else 0,
                                                    relu_in=True, p_post=p, site_post=200, rng=rng)
        dh1 = dh1.view(3, rows, Fh)
        with self._side_work(dh1, part, stack):
            for g, n in enumerate(names):
                cg = "variance_adaptor.%s_predictor.conv_layer." % n
                self._finalize_ln(part[g], nblk, 3 * Fh, cg + "conv1d_1.conv.bias")
                ops.queue_dw(self._deferred, dh1[g].view(Bn, Lp, Fh), stack[g].view(Bn, Lp, d), self._g(cg + "conv1d_1.conv.weight"), None,
                             self._acc, k=self.k_var, use_dwgemm=self._use_dwconv)
        dxin = torch.empty(3, rows, d, dtype=torch.float32, device=dev)
        ops.conv1d_dx(dh1[0].view(Bn, Lp, Fh), W1, out=dxin[0].view(Bn, Lp, d), nz1=3, sA=(rows * Fh, 0), sB=(ps, 0), sC=(rows * d, 0))
        return dxin

    def _forward(self, train, speakers, texts, src_lens, Lp, mel_lens, max_mel_len, e_targets, d_targets, pitches_raw,
                 p_control, e_control, d_control, frame_limit=None, phoneme_limit=None, defer_pred_join=False):
        """`defer_pred_join` (training): leave the predictors' stream un-joined (`_pred_fwd_pending` stays set) — for a caller that takes the
        loss in two halves (ops.fs2_loss_split: the frame-level half needs nothing of the predictors) and joins that stream itself.
        `phoneme_limit` (device int64[B], every entry = the batch's own longest text, training only): phoneme positions beyond it
        exist only because the batch was padded to a shape bucket; the predictors treat them as the zero padding the reference's
        batch has there.  `frame_limit` (device int32[1], training only): the batch was padded to a shape bucket (tts_king_amd/engine.py);
        frames t >= frame_limit[0] of every utterance do not exist in the reference's batch — the PostNet's BatchNorm statistics,
        its zero padding and (in the loss) the mel denominators are taken as if the batch ended there."""
        self.sync_shadow()
        dev, d = self.device, self.d
        Bn = texts.shape[0]
        ctx = _Ctx() if train else None
        rng = ops.rng_of(self._state()) if train else None
        p_enc, p_dec, p_var, p_post = (self.p_enc, self.p_dec, self.p_var, self.p_post) if train else (0.0, 0.0, 0.0, 0.0)
        blocks = [] if train else None
        preds = {} if train else None

        # ---- masks: fastspeech2.py:62-69.  Nothing on the device reads them (every kernel takes the lengths): when the training forward
        # has its predictor stream, they are produced there (below) instead of opening the main chain with two dependent launches
        grouped = (train and self.group_predictors and pitches_raw is not None and e_targets is not None and Lp <= self.max_seq_len)
        masks_aside = grouped and self.pred_side in ("1", "f") and mel_lens is not None and d_targets is not None and max_mel_len is not None
        src_masks = None if masks_aside else ops.length_mask(src_lens, Lp)
        # ---- encoder: phoneme embedding + position table, 4 FFT blocks: Models.py:79-112
        if Lp > self.max_seq_len:
            pe_enc = sinusoid_table(Lp, d).to(dev)        # eval-only long input: Models.py:88-99
        else:
            pe_enc = self.get("encoder.position_enc")[0]
        x = ops.gather_add(None, self._m("encoder.src_word_emb.weight"), texts, pe=pe_enc, pe_mod=Lp, rows=Bn * Lp)
        stack = torch.empty(3, Bn * Lp, d, dtype=bf16, device=dev) if grouped else None
        qkv = None
        for i in range(self.n_enc):
            nxt = "encoder.layer_stack.%d." % (i + 1) if i + 1 < self.n_enc else None
            x = self._fft_fwd("encoder.layer_stack.%d." % i, x, Bn, Lp, src_lens, self.n_head_enc, p_enc, 2 * i, rng, blocks,
                              out=stack[0] if (grouped and i == self.n_enc - 1) else None, qkv=qkv, next_pre=nxt)
            x, qkv = x if nxt else (x, None)
        # ---- variance adaptor: modules.py:142-217 (duration BEFORE the speaker embedding; energy sees the pitch embedding)
        va = "variance_adaptor."
        if grouped:
            # training with targets: the embeddings are picked by the TARGET pitch / energy, so the three predictor inputs are
            # known up front — one pass builds them (and x3), then the predictors run as grouped launches
            x3, pidx, eidx = ops.va_embed(stack, speakers, self._m("speaker_emb.weight"), Lp, pitches_raw.to(dev).float().contiguous(),
                                          self.get(va + "pitch_bins"), self._m(va + "pitch_embedding.weight"),
                                          e_targets.to(dev).float().contiguous(), self.get(va + "energy_bins"),
                                          self._m(va + "energy_embedding.weight"), row_limit=phoneme_limit)
            if self.pred_side in ("1", "f"):
                if self._pred_stream is None:
                    self._pred_stream = torch.cuda.Stream(device=dev)
                self._pred_stream.wait_stream(torch.cuda.current_stream())
                mel_masks_aside = None
                with torch.cuda.stream(self._pred_stream):
                    pred = self._predictors_fwd_grouped(stack, Bn, Lp, src_lens, p_var, rng, preds, row_limit=phoneme_limit)
                    if masks_aside:
                        T_m = min(int(max_mel_len), self.max_seq_len)
                        src_masks = ops.length_mask(src_lens, Lp)
                        mel_masks_aside = ops.length_mask(mel_lens.to(dev).long().contiguous(), T_m)
                self._pred_fwd_pending = True          # joined at the end of this forward: the loss is the first reader
            else:
                pred = self._predictors_fwd_grouped(stack, Bn, Lp, src_lens, p_var, rng, preds, row_limit=phoneme_limit)
            logd, pitch, energy = pred[0], pred[1], pred[2]
        else:
            logd = self._predictor_fwd(va + "duration_predictor.", x, Bn, Lp, src_lens, p_var, 200, rng, preds)
            x1 = ops.gather_add(x, self._m("speaker_emb.weight"), speakers, idx_div=Lp)          # fastspeech2.py:72-75
            pitch = self._predictor_fwd(va + "pitch_predictor.", x1, Bn, Lp, src_lens, p_var, 202, rng, preds)
            if pitches_raw is not None:
                pidx = ops.bucketize(pitches_raw.to(dev).float(), self.get(va + "pitch_bins"))
            else:
                pidx, pitch = ops.bucketize(pitch, self.get(va + "pitch_bins"), p_control, want_scaled=True)
            x2 = ops.gather_add(x1, self._m(va + "pitch_embedding.weight"), pidx.view(-1))
            energy = self._predictor_fwd(va + "energy_predictor.", x2, Bn, Lp, src_lens, p_var, 204, rng, preds)
            if e_targets is not None:
                eidx = ops.bucketize(e_targets.to(dev).float(), self.get(va + "energy_bins"))
            else:
                eidx, energy = ops.bucketize(energy, self.get(va + "energy_bins"), e_control, want_scaled=True)
            x3 = ops.gather_add(x2, self._m(va + "energy_embedding.weight"), eidx.view(-1))
        # ---- length regulator (+ decoder position table, fused): modules.py:196-205,225-252; Models.py:172-178
        if d_targets is not None:
            dur = d_targets.to(dev).long().contiguous()
            d_rounded = d_targets
        else:
            dur = ops.duration_round(logd, d_control)
            d_rounded = dur
        if d_targets is not None and max_mel_len is not None:
            T_full = int(max_mel_len)
        else:
            _, _, _, total = ops.length_regulator_fwd(x3.view(Bn, Lp, d), dur, 1, want_idx=False)
            T_full = max(int(total.max().item()), 1)       # data-dependent output length: one host read, as in the reference
        eval_long = (not train) and T_full > self.max_seq_len
        T = T_full if eval_long else min(T_full, self.max_seq_len)
        pe_dec = sinusoid_table(T, d).to(dev) if eval_long else self.get("decoder.position_enc")[0]
        dec_in, _, cs, mel_lens_out = ops.length_regulator_fwd(x3.view(Bn, Lp, d), dur, T, pe=pe_dec, want_idx=False)
        mlens = mel_lens.to(dev).long().contiguous() if mel_lens is not None else mel_lens_out
        if masks_aside and mel_masks_aside is not None and mel_masks_aside.shape[1] == T:
            mel_masks = mel_masks_aside
        else:
            mel_masks = ops.length_mask(mlens, T)
        # ---- decoder: Models.py:157-189
        y = dec_in.view(Bn * T, d)
        n_enc_blocks = len(blocks) if train else 0
        qkv = None
        for i in range(self.n_dec):
            nxt = "decoder.layer_stack.%d." % (i + 1) if i + 1 < self.n_dec else None
            y = self._fft_fwd("decoder.layer_stack.%d." % i, y, Bn, T, mlens, self.n_head_dec, p_dec, 100 + 2 * i, rng, blocks, qkv=qkv, next_pre=nxt)
            y, qkv = y if nxt else (y, None)
        # ---- mel_linear (fp32 output + bf16 copy for the PostNet): fastspeech2.py:102
        rows = Bn * T
        pkm = self._w1_packed.get(("lin", "mel_linear.weight")) if (self.window_ffn and self._w1_packed) else None
        if pkm is not None and y.dtype == bf16:
            mel, mel16 = ops.win_conv_dual(y.view(Bn, T, d), pkm, self.n_mel, 1, bias=self._m("mel_linear.bias"))      # window kernel, k = 1
            mel, mel16 = mel.view(rows, self.n_mel), mel16.view(rows, self.n_mel)
        else:
            mel16 = torch.empty(rows, self.n_mel, dtype=bf16, device=dev)
            mel = ops.linear(y, self._w("mel_linear.weight"), self._m("mel_linear.bias"), out_dtype=torch.float32, C2=mel16)
        fl = (frame_limit, T) if (frame_limit is not None and train) else None
        if fl is not None:
            ops.zero_frames_from(mel16, fl)          # the PostNet's first conv sees zero padding past the batch's own length
        # ---- PostNet: Layers.py:133-143, + mel: fastspeech2.py:104
        pn = []
        xin = mel16.view(Bn, T, self.n_mel)
        for i in range(5):
            pp = "postnet.convolutions.%d." % i
            # conv output stays fp32: BatchNorm divides by the batch std, which amplifies a bf16 rounding of it
            pk = self._w1_packed.get(("pn", pp + "0.conv.weight")) if (self.window_ffn and self._w1_packed) else None
            stats = None
            if pk is not None and xin.dtype == bf16:
                cw = self._table[pp + "0.conv.weight"].storage_shape
                if train and xin.shape[2] in (512, 80) and ops.bn_slab_supported(cw[0]) and self.bn_stats_in_conv:
                    # window kernel; the BatchNorm statistics partials of its output come out of its epilogue
                    yc, stats = ops.win_conv_stats(xin, pk, cw[0], cw[1], bias=self._m(pp + "0.conv.bias"), frame_limit=fl)
                else:
                    yc = ops.win_conv(xin, pk, cw[0], cw[1], bias=self._m(pp + "0.conv.bias"), out_dtype=torch.float32)   # window kernel
            else:
                yc = ops.conv1d(xin, self._w(pp + "0.conv.weight"), self._m(pp + "0.conv.bias"), out_dtype=torch.float32)
            C = yc.shape[2]
            last = i == 4
            if train and ops.bn_slab_supported(C):
                # statistics partials, then one kernel that finishes mean / rstd per channel slab and normalises
                nxt, mean, rstd, keep = ops.bn_train(yc.view(rows, C), self.get(pp + "1.running_mean"), self.get(pp + "1.running_var"),
                                                     self.get(pp + "1.num_batches_tracked").view(1), self._m(pp + "1.weight"),
                                                     self._m(pp + "1.bias"), not last, p=p_post, site=300 + i, rng=rng,
                                                     resid=mel if last else None, out_f32=last, frame_limit=fl, want_keep=True,
                                                     partials=stats)
            else:
                keep = None
                if train:
                    mean, rstd = ops.bn_train_stats(yc.view(rows, C), self.get(pp + "1.running_mean"), self.get(pp + "1.running_var"),
                                                    self.get(pp + "1.num_batches_tracked").view(1), frame_limit=fl)
                else:
                    mean, rstd = self.get(pp + "1.running_mean"), ops.rsqrt_eps(self.get(pp + "1.running_var"))
                nxt = ops.bn_apply(yc.view(rows, C), mean, rstd, self._m(pp + "1.weight"), self._m(pp + "1.bias"), not last,
                                   p=p_post, site=300 + i, rng=rng, resid=mel if last else None, out_f32=last, frame_limit=fl)
            pn.append((pp, xin, yc, mean, rstd, keep))
            xin = nxt.view(Bn, T, C) if not last else nxt
        post = xin
        if self._pred_fwd_pending and not defer_pred_join:
            torch.cuda.current_stream().wait_stream(self._pred_stream)
            self._pred_fwd_pending = False
        if train:
            ctx.blocks, ctx.preds, ctx.pn = blocks, preds, pn
            ctx.n_enc_blocks = n_enc_blocks
            ctx.dims = (Bn, Lp, T)
            ctx.texts, ctx.speakers, ctx.pidx, ctx.eidx, ctx.cs = texts, speakers, pidx, eidx, cs
            ctx.dec_out = y
            ctx.frame_limit = fl
            ctx.used = False
        out = (mel.view(Bn, T, self.n_mel), pitch, energy, logd, d_rounded, src_masks, mel_masks, mel_lens_out,
               post.view(Bn, T, self.n_mel))
        return out, ctx

    # ------------------------------------------------------------------ inference in two capturable halves
    def eval_front(self, speakers, texts, src_lens, Lp, p_control=1.0, e_control=1.0, d_control=1.0):
        """Encoder + variance adaptor of the free-running inference path (reference: fastspeech2.py:62-99 with no
        targets, modules.py:142-205) up to the per-utterance frame totals.  Only enqueues kernels (hipGraph-capturable);
        the caller reads `total.max()` on the host — the one data-dependent shape of the path — and calls `eval_back`."""
        self.sync_shadow()
        d = self.d
        Bn = texts.shape[0]
        va = "variance_adaptor."
        src_masks = ops.length_mask(src_lens, Lp)
        pe_enc = sinusoid_table(Lp, d).to(self.device) if Lp > self.max_seq_len else self.get("encoder.position_enc")[0]
        x = ops.gather_add(None, self._m("encoder.src_word_emb.weight"), texts, pe=pe_enc, pe_mod=Lp, rows=Bn * Lp)
        qkv = None
        for i in range(self.n_enc):
            nxt = "encoder.layer_stack.%d." % (i + 1) if i + 1 < self.n_enc else None
            x = self._fft_fwd("encoder.layer_stack.%d." % i, x, Bn, Lp, src_lens, self.n_head_enc, 0.0, 0, None, None, qkv=qkv, next_pre=nxt)
            x, qkv = x if nxt else (x, None)
        logd = self._predictor_fwd(va + "duration_predictor.", x, Bn, Lp, src_lens, 0.0, 0, None, None)
        x1 = ops.gather_add(x, self._m("speaker_emb.weight"), speakers, idx_div=Lp)
        pitch = self._predictor_fwd(va + "pitch_predictor.", x1, Bn, Lp, src_lens, 0.0, 0, None, None)
        pidx, pitch = ops.bucketize(pitch, self.get(va + "pitch_bins"), p_control, want_scaled=True)
        x2 = ops.gather_add(x1, self._m(va + "pitch_embedding.weight"), pidx.view(-1))
        energy = self._predictor_fwd(va + "energy_predictor.", x2, Bn, Lp, src_lens, 0.0, 0, None, None)
        eidx, energy = ops.bucketize(energy, self.get(va + "energy_bins"), e_control, want_scaled=True)
        x3 = ops.gather_add(x2, self._m(va + "energy_embedding.weight"), eidx.view(-1))
        dur = ops.duration_round(logd, d_control)
        _, _, _, total = ops.length_regulator_fwd(x3.view(Bn, Lp, d), dur, 1, want_idx=False)
        return x3, dur, total, (pitch, energy, logd, src_masks)

    def eval_back(self, x3, dur, Lp, T):
        """LengthRegulator + decoder + mel_linear + PostNet for a known frame count T (reference: modules.py:199-205,
        Models.py:157-189, fastspeech2.py:101-104).  Capturable; returns (mel, postnet mel, mel_lens, mel_masks)."""
        d = self.d
        Bn = x3.shape[0] // Lp
        dev = self.device
        pe_dec = sinusoid_table(T, d).to(dev) if T > self.max_seq_len else self.get("decoder.position_enc")[0]
        dec_in, _, _, mel_lens = ops.length_regulator_fwd(x3.view(Bn, Lp, d), dur, T, pe=pe_dec, want_idx=False)
        mel_masks = ops.length_mask(mel_lens, T)
        y = dec_in.view(Bn * T, d)
        qkv = None
        for i in range(self.n_dec):
            nxt = "decoder.layer_stack.%d." % (i + 1) if i + 1 < self.n_dec else None
            y = self._fft_fwd("decoder.layer_stack.%d." % i, y, Bn, T, mel_lens, self.n_head_dec, 0.0, 0, None, None, qkv=qkv, next_pre=nxt)
            y, qkv = y if nxt else (y, None)
        rows = Bn * T
        mel16 = torch.empty(rows, self.n_mel, dtype=bf16, device=dev)
        mel = ops.linear(y, self._w("mel_linear.weight"), self._m("mel_linear.bias"), out_dtype=torch.float32, C2=mel16)
        xin = mel16.view(Bn, T, self.n_mel)
        for i in range(5):
            pp = "postnet.convolutions.%d." % i
            pk = self._w1_packed.get(("pn", pp + "0.conv.weight")) if (self.window_ffn and self._w1_packed) else None
            if pk is not None and xin.dtype == bf16:
                cw = self._table[pp + "0.conv.weight"].storage_shape
                yc = ops.win_conv(xin, pk, cw[0], cw[1], bias=self._m(pp + "0.conv.bias"), out_dtype=torch.float32)   # window kernel
            else:
                yc = ops.conv1d(xin, self._w(pp + "0.conv.weight"), self._m(pp + "0.conv.bias"), out_dtype=torch.float32)
            C = yc.shape[2]
            last = i == 4
            nxt = ops.bn_apply(yc.view(rows, C), self.get(pp + "1.running_mean"), ops.rsqrt_eps(self.get(pp + "1.running_var")),
                               self._m(pp + "1.weight"), self._m(pp + "1.bias"), not last, resid=mel if last else None, out_f32=last)
            xin = nxt.view(Bn, T, C) if not last else nxt
        return mel.view(Bn, T, self.n_mel), xin.view(Bn, T, self.n_mel), mel_lens, mel_masks

    # ------------------------------------------------------------------ backward
    def _backward_from_autograd(self, ctx, dmel, dpost, dpitch, denergy, dlogd):
        dev = self.device
        Bn, Lp, T = ctx.dims

        def f32(g, shape):
            if g is None:
                return torch.zeros(shape, dtype=torch.float32, device=dev)
            return g.to(torch.float32).contiguous()
        dmel, dpost = f32(dmel, (Bn, T, self.n_mel)), f32(dpost, (Bn, T, self.n_mel))
        with torch.no_grad():
            dmel_sum = ops.add_f32(dmel, dpost)
            self.backward_native(ctx, dmel_sum, dpost, f32(dpitch, (Bn, Lp)), f32(denergy, (Bn, Lp)), f32(dlogd, (Bn, Lp)))
            # one dropout-counter tick per micro-step, as main_train_step / graph.make_enqueue do: Philox masks are a
            # function of (seed, step, site, element), so without it every `loss.backward()` step would reuse one mask
            ops.rng_advance(self._state())

    def _finalize_ln(self, partials, nblk, ncol, first_key):
        """partials [nblk][ncol] = dbias | dgamma | dbeta (| dhead_w | dhead_b) -> the flat gradient buffer, where the
        sub-layer bias, LayerNorm weight and LayerNorm bias (and the predictor head) sit back to back in that order
        starting at `first_key` (params.py builds them so): ONE finalize launch per LayerNorm."""
        off = self._table[first_key].offset
        ops.colsum_finalize(partials, nblk, ncol, ncol, self._flat_grad[off:off + ncol], accumulate=self._acc, defer=self._deferred_fin)

    class _SideWork:
        """Parameter-gradient work (dW GEMMs, bias / LayerNorm column sums) runs on a second HIP stream: nothing on the
        dX critical path reads it, and most backward kernels fill a fraction of the 256 CUs, so the two chains overlap
        (captured as parallel branches of the step's hipGraph).  Entering makes the side stream wait for everything the
        main stream has enqueued (the producers of the tensors about to be read); leaving marks those tensors as in use
        on the side stream so the caching allocator does not hand their memory to a later main-stream kernel."""

        def __init__(self, model, tensors):
            self.m, self.tensors = model, tensors

        def __enter__(self):
            m = self.m
            if m._side is None:
                return self
            m._side.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(m._side)
            self.ctx.__enter__()
            return self

        def __exit__(self, *exc):
            m = self.m
            if m._side is None:
                return False
            for t in self.tensors:
                if t is not None:
                    t.record_stream(m._side)
            self.ctx.__exit__(*exc)
            return False

    def _side_work(self, *tensors):
        return FastSpeech2._SideWork(self, tensors)

    def _join_side(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def _raw_out_mode(self):
        """How a block hands its input gradient to the block in front of it (see _fft_bwd): "pre" = not at all — the consumer's first
        backward kernel (ttsk_layernorm_bwd_proj) computes it from dqkv; True = split-K slabs; the consumer must be that kernel."""
        if (self.fused_qkv_dx and self.fused_ln_bwd and self.raw_slabs and self.window_ffn and self._w1_packed and self.d == 256 and
                self.k2 == 1 and self.d_ff == 1024):
            return "pre"
        return True

    def _fft_bwd(self, saved, dx2, rng, raw_out=False):
        """Backward of one FFTBlock.  `dx2`: gradient of the block output — a bf16 tensor, or (Slabs, residual) when the dX GEMM
        that produced it left its split-K partial tiles un-reduced (the next block's `raw_out`): the LayerNorm backward sums them
        while it reads its rows, so no reducer launch and no bf16 copy of that gradient exist.  Returns the gradient of the block
        input in the same two forms."""
        (pre, x, qkv, probs, o, z1, mean1, rstd1, x1, h, z2, mean2, rstd2, Bn, S, lens, H, p, site, o32) = saved
        d, rows = self.d, Bn * S
        dk = d // H
        flash = probs is not None and probs.dim() == 2       # forward was the flash kernel: `probs` holds the row LSE
        Sp = 0 if flash else probs.shape[2]
        a, f = pre + "slf_attn.", pre + "pos_ffn."
        dev = z2.device
        # ---- FFN tail: LN backward (PAD rows carry no gradient), dropout mask regenerated
        pk2 = self._w1_packed.get(("w2T", f + "w_2.weight")) if (self.window_ffn and self._w1_packed) else None
        pre2 = None
        if isinstance(dx2, tuple) and len(dx2) == 3:          # (dqkv, packed transposed q|k|v weight, residual) of the block behind
            pre2, sl2, r2, dd2 = (dx2[0], dx2[1]), None, dx2[2], None
        else:
            sl2, r2, dd2 = (dx2[0], dx2[1], None) if isinstance(dx2, tuple) else (None, None, dx2)
        fuse = self.fused_ln_bwd and d == 256 and self.k2 == 1
        dh = None
        if fuse and pk2 is not None and h.shape[-1] == 1024:
            # (the q|k|v input gradient of the block behind +) LN backward + w_2's dX (ReLU gate on the way out) in one launch
            dz2, dy2, part, nblk, dh = ops.layernorm_bwd_proj(dd2, z2, mean2, rstd2, self._m(f + "layer_norm.weight"), pk2, h.shape[-1], lens, S,
                                                              p_pre=p, site_pre=site + 1, rng=rng, slabs=sl2, R=r2, gate=h, pre=pre2)
            dh = dh.view(Bn, S, -1)
        else:
            dz2, dy2, part, nblk = ops.layernorm_bwd(dd2, z2, mean2, rstd2, self._m(f + "layer_norm.weight"), self._m(f + "layer_norm.bias"),
                                                     lens, S, p_pre=p, site_pre=site + 1, rng=rng, slabs=sl2, R=r2)
        # ---- w_2 (k=1): dW, dX gated by the ReLU
        with self._side_work(dy2, part, h):
            self._finalize_ln(part, nblk, 3 * d, f + "w_2.bias")
            ops.queue_dw(self._deferred, dy2.view(Bn, S, d), h, self._g(f + "w_2.weight"), lens, self._acc, k=self.k2, use_dwgemm=self._use_dwconv)
        if dh is not None:
            pass
        elif pk2 is not None and self.k2 == 1:
            dh = ops.win_conv(dy2.view(Bn, S, d), pk2, h.shape[-1], 1, gate=h)         # dX as a forward conv on the transposed pack, ReLU gate on the way out
        else:
            dh = ops.conv1d_dx(dy2.view(Bn, S, d), self._w(f + "w_2.weight"), G=h)
        # ---- w_1 (k=9): bias, dW, dX + residual gradient; the dX stays in split-K form for the attention LayerNorm's backward
        with self._side_work(dh, x1):
            ops.colsum_into(dh.view(rows, -1), self._g(f + "w_1.bias"), defer=self._deferred_fin, accumulate=self._acc)
            if self._use_dwconv and self._deferred.group is not None and Bn <= 64 and dh.dtype == bf16 and ops.dwconv_supported(dh.shape[-1], d, self.k1):
                # the taps share one fetch of dh and one window of x1 (csrc/dwconv.hip); dh is zero at PAD rows (the LayerNorm backward
                # that produced dy2 gives them no gradient), so only the rows of each utterance's own length are walked
                self._deferred.dwconv.append((dh, x1.view(Bn, S, d), self._g(f + "w_1.weight"), lens, self._acc))
            else:
                ops.conv1d_dw(dh, x1.view(Bn, S, d), self._g(f + "w_1.weight"), k=self.k1, defer=self._deferred, accumulate=self._acc)
        # ---- attention tail
        do = delta = None
        if self.raw_slabs:
            pk1 = self._w1_packed.get(("w1T", f + "w_1.weight")) if (self.window_ffn and self._w1_packed) else None
            if pk1 is not None:
                sl = ops.win_conv_split(dh, pk1, d, self.k1)       # four 256-channel slices of the 1024-channel contraction, one launch
            else:
                sl = ops.conv1d_dx(dh, self._w(f + "w_1.weight"), raw=True)
            pkf = self._w1_packed.get(("fcT", a + "fc.weight")) if (self.window_ffn and self._w1_packed) else None
            if fuse and pkf is not None and lens is not None:
                # LN backward + fc's dX (+ the attention backward's delta) in one launch
                if flash and o32 is not None:
                    delta = torch.empty(Bn * H, S, dtype=torch.float32, device=dev)
                dz1, dy1, part, nblk, do = ops.layernorm_bwd_proj(None, z1, mean1, rstd1, self._m(a + "layer_norm.weight"), pkf, d, lens, S,
                                                                  p_pre=p, site_pre=site, rng=rng, slabs=sl, R=dz2,
                                                                  delta_o32=o32 if delta is not None else None, delta_out=delta)
            else:
                dz1, dy1, part, nblk = ops.layernorm_bwd(None, z1, mean1, rstd1, self._m(a + "layer_norm.weight"), self._m(a + "layer_norm.bias"),
                                                         lens, S, p_pre=p, site_pre=site, rng=rng, slabs=sl, R=dz2)
        else:
            dx1 = ops.conv1d_dx(dh, self._w(f + "w_1.weight"), R=dz2.view(Bn, S, d))
            dz1, dy1, part, nblk = ops.layernorm_bwd(dx1.view(rows, d), z1, mean1, rstd1, self._m(a + "layer_norm.weight"),
                                                     self._m(a + "layer_norm.bias"), lens, S, p_pre=p, site_pre=site, rng=rng)
        with self._side_work(dy1, part, o):
            self._finalize_ln(part, nblk, 3 * d, a + "fc.bias")
            ops.queue_dw(self._deferred, dy1.view(Bn, S, d), o.view(Bn, S, d), self._g(a + "fc.weight"), lens, self._acc, use_dwgemm=self._use_dwconv)
        pkf = self._w1_packed.get(("fcT", a + "fc.weight")) if (self.window_ffn and self._w1_packed) else None
        if do is not None:
            pass
        elif pkf is not None:
            if flash and o32 is not None:
                # the attention backward's delta = rowsum(dO o O) per head, written by this conv's epilogue while it stores dO
                delta = torch.empty(Bn * H, S, dtype=torch.float32, device=dev)
            do = ops.win_conv(dy1.view(Bn, S, d), pkf, d, 1, delta_o32=o32 if delta is not None else None, delta_out=delta).view(rows, d)
        else:
            do = ops.linear_dx(dy1, self._w(a + "fc.weight"))
        # ---- attention core: dP = dO V^T ; dS = softmax'(P, dP)/sqrt(dk) ; dQ = dS K ; dK = dS^T Q ; dV = P^T dO
        if flash:
            dqkv = ops.flash_attention_bwd(qkv, o, do, probs, lens, Bn, H, S, o32=o32, delta=delta)      # P recomputed per tile; dQ, dK, dV in two launches
        else:
            dqkv = torch.empty(rows, 3 * d, dtype=bf16, device=dev)
            dP = torch.empty(Bn * H, S, Sp, dtype=torch.float32, device=dev)
            ops.gemm(do, qkv[:, 2 * d:], dP, S, S, dk, d, 3 * d, Sp, nz1=Bn, nz2=H, sA=(S * d, dk), sB=(S * 3 * d, dk),
                     sC=(H * S * Sp, S * Sp))
            dS = ops.softmax_bwd(probs, dP, dk ** -0.5)
            ops.gemm(dS, qkv[:, d:], dqkv, S, dk, S, Sp, 3 * d, 3 * d, flags=ops.B_TR, nz1=Bn, nz2=H,
                     sA=(H * S * Sp, S * Sp), sB=(S * 3 * d, dk), sC=(S * 3 * d, dk))
            kv = ops.GemmGroup()         # dK and dV are independent: one grouped launch
            ops.gemm(dS, qkv, dqkv[:, d:], S, dk, S, Sp, 3 * d, 3 * d, flags=ops.A_TR | ops.B_TR, nz1=Bn, nz2=H,
                     sA=(H * S * Sp, S * Sp), sB=(S * 3 * d, dk), sC=(S * 3 * d, dk), group=kv)
            ops.gemm(probs, do, dqkv[:, 2 * d:], S, dk, S, Sp, d, 3 * d, flags=ops.A_TR | ops.B_TR, nz1=Bn, nz2=H,
                     sA=(H * S * Sp, S * Sp), sB=(S * d, dk), sC=(S * 3 * d, dk), group=kv)
            kv.flush()
        # ---- q|k|v projections
        with self._side_work(dqkv, x):
            ops.colsum_into(dqkv, self._g(a + "w_qs.bias", 3 * d), defer=self._deferred_fin, accumulate=self._acc)
            ops.queue_dw(self._deferred, dqkv.view(Bn, S, 3 * d), x.view(Bn, S, d), self._g(a + "w_qs.weight", 3 * d * d).view(3 * d, d), lens,
                         self._acc, use_dwgemm=self._use_dwconv)
        if raw_out and self.raw_slabs:
            pkq = self._w1_packed.get(("qkvT", a + "w_qs.weight")) if (self.window_ffn and self._w1_packed) else None
            if pkq is not None and raw_out == "pre" and d == 256:
                return (dqkv, pkq, dz1)          # the consumer (the next block's first backward kernel) multiplies by the weight itself
            if pkq is not None:
                return (ops.win_conv_split(dqkv.view(Bn, S, 3 * d), pkq, d, 1), dz1)
            return (ops.linear_dx(dqkv, self._w(a + "w_qs.weight", 3 * d), raw=True), dz1)
        pkq = self._w1_packed.get(("qkvT", a + "w_qs.weight")) if (self.window_ffn and self._w1_packed and self.fused_qkv_dx) else None
        if pkq is not None and d == 256 and dqkv.shape[1] == 768:
            return ops.qkv_dx(dqkv, pkq, R=dz1)       # the first block of the stack: the same 32-row product as a kernel of its own
        return ops.linear_dx(dqkv, self._w(a + "w_qs.weight", 3 * d), R=dz1)

    def _predictor_bwd(self, pre, saved, dout, rng, R):
        (x, h1, m1, r1, a1, h2, m2, r2, Bn, Lp, lens, p, site) = saved
        d, rows = self.d, Bn * Lp
        c = pre + "conv_layer."
        Fh = h1.shape[-1]
        hw = self._m(pre + "linear_layer.weight").view(-1)
        dh2, _, part, nblk = ops.layernorm_bwd(None, h2.view(rows, Fh), m2, r2, self._m(c + "layer_norm_2.weight"),
                                               self._m(c + "layer_norm_2.bias"), lens, Lp, relu_in=True, p_post=p,
                                               site_post=site + 1, rng=rng, dhead=dout.contiguous().view(-1), head_w=hw)
        with self._side_work(dh2, part, a1):
            self._finalize_ln(part, nblk, 4 * Fh + 1, c + "conv1d_2.conv.bias")
            ops.conv1d_dw(dh2.view(Bn, Lp, Fh), a1.view(Bn, Lp, Fh), self._g(c + "conv1d_2.conv.weight"), k=self.k_var, defer=self._deferred, accumulate=self._acc)
        da1 = ops.conv1d_dx(dh2.view(Bn, Lp, Fh), self._w(c + "conv1d_2.conv.weight"))
        dh1, _, part, nblk = ops.layernorm_bwd(da1.view(rows, Fh), h1.view(rows, Fh), m1, r1, self._m(c + "layer_norm_1.weight"),
                                               self._m(c + "layer_norm_1.bias"), None, 0, relu_in=True, p_post=p,
                                               site_post=site, rng=rng)
        with self._side_work(dh1, part, x):
            self._finalize_ln(part, nblk, 3 * Fh, c + "conv1d_1.conv.bias")
            ops.conv1d_dw(dh1.view(Bn, Lp, Fh), x.view(Bn, Lp, d), self._g(c + "conv1d_1.conv.weight"), k=self.k_var, defer=self._deferred, accumulate=self._acc)
        return ops.conv1d_dx(dh1.view(Bn, Lp, Fh), self._w(c + "conv1d_1.conv.weight"), R=R)

    def _finalize_loss(self):
        if self._loss_finalize is not None:
            fin, self._loss_finalize = self._loss_finalize, None
            fin()

    def abort_step(self):
        """Forget the stream bookkeeping of a step whose Python ran only in part (a hipGraph capture of it was aborted)."""
        self._ctx = None
        self._dw_side_pending = False
        self._pred_fwd_pending = False
        self._var_on_pred, self._loss_finalize = False, None

    def backward_group_order(self):
        """The parameter groups in the order backward_native completes their gradients (= reverse forward order; the flat
        gradient buffer is laid out in forward order, so everything at or above a finished group's offset is final)."""
        return (["postnet", "mel_linear"] + ["decoder.%d" % i for i in range(self.n_dec - 1, -1, -1)] + ["variance_adaptor"] +
                ["encoder.%d" % i for i in range(self.n_enc - 1, -1, -1)] + ["embedding"])

    def _launch_dw_side(self, everything=False):
        """The weight-gradient work queued so far (PostNet, mel_linear, decoder, predictors), after the decoder's backward, on the second
        stream beside the encoder-side dX chain on the main stream: w_1's six gradients (dwconv, 192 workgroups = one per CU on 3/4 of
        the chip), then the other 256-multiple ones (dwgemm, grid capped at `dw_side_wgs`) with their slab reducer.  The column sums and
        the few 128 x 128-tile grouped problems (80-channel outputs) wait for the final phase (measured: beside dwconv / dwgemm their HBM
        traffic slows the whole step).
        `everything` (the data-parallel "side" schedule): nothing may stay queued — neither the share `dw_side_frac` leaves for the
        final flush nor those grouped problems: the caller reduces every split-K slab and announces the buckets right behind this."""
        if self._dw_side is None:
            self._dw_side = torch.cuda.Stream(device=self.device)
        self._dw_side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._dw_side):
            ops.stamp("side.start")
            ops.flush_deferred_gemms(self._deferred, max_wgs=self.dw_side_wgs, frac=1.0 if everything else self.dw_side_frac,
                                     small_too=True if everything else self.side_small)
            ops.stamp("side.end")
        self._dw_side_pending = True

    def _mark_bucket(self, name):
        """The notifier's `mark` in the "side" / "late" data-parallel schedules: group `name` is complete once everything queued so far
        has been flushed — remember where the GEMM and reducer queues stand."""
        self._dp_marks.append((name, len(self._deferred.group), len(self._deferred), len(self._deferred.dwconv)))

    def _launch_dw_side_buckets(self, on_bucket, ready):
        """Data-parallel "side" schedule, after the decoder's backward: exactly the single-GPU schedule's second-stream work (`_launch_dw_side`:
        w_1's gradients on dwconv, the other 256-multiple ones on dwgemm with their slab reducer, the few grouped problems, the column
        sums — everything queued so far, so every PostNet / mel_linear / decoder gradient is final when it ends), and behind it, FROM THAT
        STREAM, the all-reduce of every bucket those groups complete — beside the encoder-side dX chain on the main stream, ahead of the
        final flush, which announces the encoder-side groups.  (Round 3 first cut this work at the bucket boundaries, a launch set per
        bucket: with dwconv / dwgemm covering all six blocks in one launch each that only added launches: 3.47 vs 3.09 ms on one GPU.)"""
        self._launch_dw_side(everything=True)
        q = self._deferred
        if q.group or q.dwconv or q.dwgemm:
            # a problem still queued would write its gradient (or its split-K slabs) AFTER the reduce / all-reduce below read them
            raise RuntimeError("data-parallel side schedule: %d grouped / %d dwconv / %d dwgemm weight-gradient problems still queued "
                               "before the buckets are announced" % (len(q.group or ()), len(q.dwconv), len(q.dwgemm)))
        # the split-K slabs of the grouped problems just launched: summed here, not with the final flush (the buckets must be final)
        marks, self._dp_marks = self._dp_marks, []
        with torch.cuda.stream(self._dw_side):
            ops.flush_deferred_prefix(self._deferred, 0, len(self._deferred))
            if self._deferred_fin:                                  # the column sums queued so far belong to these buckets too
                self._dp_keep = (self._dp_keep or []) + [k for _, k in self._deferred_fin]
                ops.flush_finalize(self._deferred_fin)
            for name, _, _, _ in marks:
                on_bucket(name)               # groups in completion order: a bucket goes out when its lowest group has been announced
        self._dw_side_pending = True

    def _flush_param_grads(self):
        """Run the queued weight-gradient work (grouped dW GEMMs, split-K reducers, column sums, scatter-sums)."""
        self._join_side()
        if self._dw_side_pending:
            # The main stream gets here ~0.1 ms before the capped dW group on the side stream ends.  The column sums / scatter-sums read
            # only activations and gradients the main chain produced: they run now, beside the side stream; the encoder-side dW group
            # queues behind the first one on the side stream; the join comes last, before the split-K reducer.
            cur = torch.cuda.current_stream()
            launch = ops.upload_deferred_gemms(self._deferred, max_wgs=0, with_dwconv=False, with_dwgemm=False)     # the tables now, on this stream
            # Three branches from here (a replayed graph runs at most three queues side by side): the encoder blocks' w_1 gradients
            # (dwconv, 128 workgroups) and behind them the few grouped problems (80-channel outputs) on one stream; the encoder side's
            # dwgemm problems on the side stream (behind the decoder's work there); the column sums on this one.
            if self._fin_side is None:
                self._fin_side = torch.cuda.Stream(device=self.device)
            self._fin_side.wait_stream(cur)
            with torch.cuda.stream(self._fin_side):
                ops.flush_dwconv(self._deferred)
                ops.stamp("fin.dwconv")
                launch()
                ops.stamp("fin.small")
            self._fin_pending = True
            self._dw_side.wait_stream(cur)
            with torch.cuda.stream(self._dw_side):
                ops.flush_dwgemm(self._deferred)
                ops.stamp("fin.dwgemm")
            ops.flush_finalize(self._deferred_fin)
            ops.stamp("fin.colsum")
            self._finalize_loss()       # the two-stream loss's values: behind the column sums, on the branch of the final phase that ends first
            cur.wait_stream(self._dw_side)
            if getattr(self, "_fin_pending", False):
                cur.wait_stream(self._fin_side)
                self._fin_pending = False
            self._dw_side_pending = False
        ops.flush_deferred(self._deferred)
        ops.flush_finalize(self._deferred_fin)
        self._dp_keep = None

    @staticmethod
    def _stack3(dlogd, dpitch, denergy):
        """(3, B, L) fp32 = (dlogd, dpitch, denergy).  ops.fs2_loss hands them out as the three slices of one buffer (no copy);
        gradients that arrive separately (the autograd bridge) are copied into one."""
        base = dlogd._base
        n = dlogd.numel()
        if (base is not None and dpitch._base is base and denergy._base is base and base.dim() == 3 and base.shape[0] == 3 and
                base.is_contiguous() and dlogd.data_ptr() == base.data_ptr() and dpitch.data_ptr() == base.data_ptr() + 4 * n and
                denergy.data_ptr() == base.data_ptr() + 8 * n):
            return base
        out = torch.empty((3,) + tuple(dlogd.shape), dtype=torch.float32, device=dlogd.device)
        for i, t in enumerate((dlogd, dpitch, denergy)):
            out[i].copy_(t)               # a device-to-device copy, no arithmetic
        return out

    def backward_native(self, ctx, dmel_sum, dpost, dpitch, denergy, dlogd, on_bucket=None, accumulate=None):
        """d(loss)/d(params) into the flat gradient buffer.
        dmel_sum = dL/dmel (direct terms) + dL/dpost, dpost = dL/dpost — fp32 (B,T,n_mel); dpitch/denergy/dlogd fp32 (B,L).
        `on_bucket(name)` is called when the gradients of a top-level group are complete (data-parallel overlap).
        `accumulate`: True adds to what the buffer holds (the reference's `.grad +=`, train.py:43-44, micro-steps 2.. of a
        `grad_acc_step` cycle); False OVERWRITES every gradient element (the first micro-step after an update: the buffer need not be
        zero, so the optimizer's kernel need not zero it — 4 B per parameter less on the step's serial tail, and no read of the old
        value in the weight-gradient epilogues); None: whatever `grads_partial` says the buffer holds."""
        if ctx is None or ctx.used:
            raise RuntimeError("backward called without a matching training forward")
        ctx.used = True
        self._acc = bool(self.grads_partial if accumulate is None else accumulate)
        self.grads_partial = True
        rng = ops.rng_of(self._state())
        Bn, Lp, T = ctx.dims
        d, rows, nm = self.d, Bn * T, self.n_mel
        # split-K slabs of the weight-gradient GEMMs are summed by ONE batched reducer launch per parameter group when a
        # data-parallel reducer is waiting for finished buckets, otherwise once at the end (only Adam reads them)
        self._deferred = ops.DeferQueue(group_gemms=self.group_param_grads)
        self._deferred_fin = []
        if self.overlap_param_grads and self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        if self.dp_schedule not in ("side", "late"):
            raise ValueError("dp_schedule must be 'side' or 'late' (got %r)" % (self.dp_schedule,))
        dp_side = on_bucket is not None and self.dw_side_wgs > 0 and self.group_param_grads and not self.overlap_param_grads
        self._dp_marks = []
        # w_1's weight gradients on the tap-sharing kernel: one launch for the six decoder blocks (192 workgroups) and one for the encoder's;
        # not when buckets are flushed one by one (no second stream): those flushes would launch them two at a time
        self._use_dwconv = self.dwconv and (on_bucket is None or dp_side)
        notifier = _GroupNotifier(self.backward_group_order(), on_bucket, self._flush_param_grads, mark=self._mark_bucket if dp_side else None)
        notify = notifier.done
        ops.stamp("bwd.start")
        # ---- the predictors' backward up to their input gradients, on a stream of its own beside the PostNet's (see pred_side); not in
        # the schedules that flush the deferred queue before the decoder is done
        pred_dxin = None
        if (self.pred_side in ("1", "b") and "grouped" in ctx.preds and (on_bucket is None or dp_side) and not self.overlap_param_grads
                and self.group_param_grads):
            if self._pred_stream is None:
                self._pred_stream = torch.cuda.Stream(device=self.device)
            if not self._var_on_pred:             # (the two-stream loss left dlogd / dpitch / denergy ON that stream: nothing of this one is needed)
                self._pred_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._pred_stream):
                pred_dxin = self._predictors_bwd_inputs(ctx.preds["grouped"], self._stack3(dlogd, dpitch, denergy), rng)
        elif self._var_on_pred:                   # the predictors' backward runs on this stream: it needs the two-stream loss's other half
            torch.cuda.current_stream().wait_stream(self._pred_stream)
            self._finalize_loss()
        self._var_on_pred = False
        # ---- PostNet (last layer first)
        dout = dpost.view(rows, nm)
        bn_partials = None          # BatchNorm-backward statistics of layer i, when conv i+1's input-gradient kernel emitted them
        for i in range(4, -1, -1):
            pp, xin, yc, mean, rstd, keep = ctx.pn[i]
            C = yc.shape[2]
            dy = ops.bn_bwd(dout, yc.view(rows, C), mean, rstd, self._m(pp + "1.weight"), self._m(pp + "1.bias"), i < 4,
                            p=self.p_post, site=300 + i, rng=rng, dgamma=self._g(pp + "1.weight"), dbeta=self._g(pp + "1.bias"),
                            frame_limit=ctx.frame_limit, keep=keep, accumulate=self._acc, partials=bn_partials)
            bn_partials = None
            with self._side_work(dy, xin):
                ops.colsum_into(dy, self._g(pp + "0.conv.bias"), defer=self._deferred_fin, accumulate=self._acc)
                # (no `lens`: the PostNet's BatchNorm runs over the PAD rows too, Layers.py:133-143 — its gradients there are not zero)
                ops.queue_dw(self._deferred, dy.view(Bn, T, C), xin, self._g(pp + "0.conv.weight"), None, self._acc, k=5,
                             use_dwgemm=self._use_dwconv)
            pkt = self._w1_packed.get(("pnT", pp + "0.conv.weight")) if (self.window_ffn and self._w1_packed) else None
            if i > 0 and pkt is not None:
                cw = self._table[pp + "0.conv.weight"].storage_shape
                pb, _, ycb, meanb, rstdb, keepb = ctx.pn[i - 1]
                if (self.bn_bwd_stats_in_conv and C in (512, 80) and cw[2] == 512 and ycb.dtype == torch.float32 and ops.bn_slab_supported(cw[2])
                        and (keepb is not None or self.p_post == 0.0)):
                    # ... which also sums the BatchNorm-backward statistics of the layer below over its output tile
                    dout, bn_partials = ops.win_conv_bnb(dy.view(Bn, T, C), pkt, cw[2], cw[1], ycb.view(rows, cw[2]), meanb, rstdb,
                                                         self._m(pb + "1.weight"), self._m(pb + "1.bias"), True, p=self.p_post, keep=keepb,
                                                         frame_limit=ctx.frame_limit)
                    dout = dout.view(rows, -1)
                else:
                    dout = ops.win_conv(dy.view(Bn, T, C), pkt, cw[2], cw[1]).view(rows, -1)   # dX as a forward conv on the transposed pack
            elif i > 0:
                dout = ops.conv1d_dx(dy.view(Bn, T, C), self._w(pp + "0.conv.weight")).view(rows, -1)
            else:
                if pkt is not None:          # the first conv's input gradient (512 -> 80) + the mel terms' own gradient, on the window kernel
                    cw = self._table[pp + "0.conv.weight"].storage_shape
                    dmel_tot = ops.win_conv_resid(dy.view(Bn, T, C), pkt, dmel_sum.view(Bn, T, nm), cw[2], cw[1]).view(rows, nm)
                else:
                    dmel_tot = ops.conv1d_dx(dy.view(Bn, T, C), self._w(pp + "0.conv.weight"), R=dmel_sum.view(Bn, T, nm)).view(rows, nm)
                if ctx.frame_limit is not None:
                    ops.zero_frames_from(dmel_tot, ctx.frame_limit)      # the conv's reach past the batch's own length is not a frame
        notify("postnet")
        ops.stamp("bwd.postnet_done")
        # ---- mel_linear
        with self._side_work(dmel_tot, ctx.dec_out):
            ops.colsum_into(dmel_tot, self._g("mel_linear.bias"), defer=self._deferred_fin, accumulate=self._acc)
            ops.linear_dw(dmel_tot, ctx.dec_out, self._g("mel_linear.weight"), defer=self._deferred, accumulate=self._acc)
        pkm = self._w1_packed.get(("linT", "mel_linear.weight")) if (self.window_ffn and self._w1_packed) else None
        if pkm is not None and dmel_tot.dtype == bf16:
            dx = ops.win_conv(dmel_tot.view(Bn, T, nm), pkm, d, 1).view(rows, d)      # mel_linear's input gradient: a k = 1 conv on the transposed pack
        else:
            dx = ops.linear_dx(dmel_tot, self._w("mel_linear.weight"))
        notify("mel_linear")
        # ---- decoder
        for i in range(self.n_dec - 1, -1, -1):
            dx = self._fft_bwd(ctx.blocks[ctx.n_enc_blocks + i], dx, rng, raw_out=self._raw_out_mode() if i > 0 else False)
            notify("decoder.%d" % i)
        ops.stamp("bwd.decoder_done")
        if pred_dxin is not None:
            torch.cuda.current_stream().wait_stream(self._pred_stream)    # long done; its queued dW work joins the second stream's
        if dp_side and self.dp_schedule == "side":
            self._launch_dw_side_buckets(on_bucket, notifier.ready)
        elif self.dw_side_wgs > 0 and (on_bucket is None or dp_side) and self.group_param_grads and not self.overlap_param_grads:
            self._launch_dw_side()
        # ---- length regulator: segment sums (the position table has no parameters)
        dx3 = ops.length_regulator_bwd(dx.view(Bn, T, d), ctx.cs, Lp).view(Bn * Lp, d)
        # ---- variance adaptor, reverse order of modules.py:158-193
        va = "variance_adaptor."
        with self._side_work(dx3):
            ops.scatter_sum(dx3, ctx.eidx.view(-1), self._g(va + "energy_embedding.weight"), defer=self._deferred_fin, accumulate=self._acc)
        if "grouped" in ctx.preds:
            dx2, dx1, dxe = self._predictors_bwd_grouped(ctx.preds["grouped"], None if pred_dxin is not None else
                                                         self._stack3(dlogd, dpitch, denergy), rng, dx3, dxin=pred_dxin)
            with self._side_work(dx2, dx1):
                ops.scatter_sum(dx2, ctx.pidx.view(-1), self._g(va + "pitch_embedding.weight"), defer=self._deferred_fin, accumulate=self._acc)
                ops.scatter_sum(dx1, ctx.speakers, self._g("speaker_emb.weight"), idx_div=Lp, defer=self._deferred_fin, accumulate=self._acc)
        else:
            dx2 = self._predictor_bwd(va + "energy_predictor.", ctx.preds[va + "energy_predictor."], denergy, rng, dx3.view(Bn, Lp, d))
            with self._side_work(dx2):
                ops.scatter_sum(dx2.view(Bn * Lp, d), ctx.pidx.view(-1), self._g(va + "pitch_embedding.weight"), defer=self._deferred_fin, accumulate=self._acc)
            dx1 = self._predictor_bwd(va + "pitch_predictor.", ctx.preds[va + "pitch_predictor."], dpitch, rng, dx2)
            with self._side_work(dx1):
                ops.scatter_sum(dx1.view(Bn * Lp, d), ctx.speakers, self._g("speaker_emb.weight"), idx_div=Lp, defer=self._deferred_fin, accumulate=self._acc)
            dxe = self._predictor_bwd(va + "duration_predictor.", ctx.preds[va + "duration_predictor."], dlogd, rng, dx1)
        notify("variance_adaptor")
        # ---- encoder
        dx = dxe.view(Bn * Lp, d)
        for i in range(self.n_enc - 1, -1, -1):
            dx = self._fft_bwd(ctx.blocks[i], dx, rng, raw_out=self._raw_out_mode() if i > 0 else False)
            notify("encoder.%d" % i)
        with self._side_work(dx):
            ops.scatter_sum(dx, ctx.texts.view(-1), self._g("encoder.src_word_emb.weight"), skip_row=0, defer=self._deferred_fin, accumulate=self._acc)   # padding_idx=0
        notify("embedding")
        ops.stamp("bwd.chain_done")
        self._flush_param_grads()
        ops.stamp("bwd.flushed")
        if dp_side:
            for name, _, _, _ in self._dp_marks:    # everything is in the buffer now: the remaining buckets, in completion order
                on_bucket(name)
            self._dp_marks = []
        self._finalize_loss()       # (a schedule without the second stream: here, where this stream has seen both halves of the loss)
        self._ctx = None
