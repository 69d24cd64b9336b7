"""Parameter inventory of FastSpeech2 with the reference's `state_dict` key names, and the flat storage
that backs it.

All trainable parameters live in ONE fp32 buffer (`flat`), their gradients in a second one of the same layout
(`flat_grad`) and their bf16 copies, which the MFMA kernels read, in a third (`shadow`).  Consequences:
gradient clipping is one norm over one buffer, Adam is one kernel launch, the data-parallel all-reduce works on
contiguous buckets, and the bf16 shadow is refreshed by the Adam kernel itself.

Conv1d weights are stored tap-major, (Cout, k, Cin), because the implicit-GEMM kernel wants the input channels
contiguous per tap; the `nn.Parameter` users see is a permuted VIEW with the reference's shape (Cout, Cin, k),
so `state_dict()` / `load_state_dict()` round-trip reference checkpoints without copies.
"""
from collections import OrderedDict

TRAIN, FROZEN, UNUSED, BUFFER = "train", "frozen", "unused", "buffer"


class Entry:
    __slots__ = ("key", "shape", "kind", "conv", "offset", "dtype")

    def __init__(self, key, shape, kind, conv=False, dtype="float32"):
        self.key, self.shape, self.kind, self.conv, self.dtype = key, tuple(shape), kind, conv, dtype
        self.offset = -1

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def storage_shape(self):
        """(Cout, k, Cin) for conv weights, the reference shape otherwise."""
        if self.conv:
            co, ci, k = self.shape
            return (co, k, ci)
        return self.shape


def _fft_block(pre, d, d_ff, k1, k2):
    a, f = pre + "slf_attn.", pre + "pos_ffn."
    return [
        Entry(a + "w_qs.weight", (d, d), TRAIN), Entry(a + "w_ks.weight", (d, d), TRAIN), Entry(a + "w_vs.weight", (d, d), TRAIN),
        Entry(a + "w_qs.bias", (d,), TRAIN), Entry(a + "w_ks.bias", (d,), TRAIN), Entry(a + "w_vs.bias", (d,), TRAIN),
        Entry(a + "fc.weight", (d, d), TRAIN), Entry(a + "fc.bias", (d,), TRAIN),
        Entry(a + "layer_norm.weight", (d,), TRAIN), Entry(a + "layer_norm.bias", (d,), TRAIN),
        Entry(f + "w_1.weight", (d_ff, d, k1), TRAIN, conv=True), Entry(f + "w_1.bias", (d_ff,), TRAIN),
        Entry(f + "w_2.weight", (d, d_ff, k2), TRAIN, conv=True), Entry(f + "w_2.bias", (d,), TRAIN),
        Entry(f + "layer_norm.weight", (d,), TRAIN), Entry(f + "layer_norm.bias", (d,), TRAIN),
    ]


def _predictor(pre, d, filt, k):
    c = pre + "conv_layer."
    return [
        Entry(c + "conv1d_1.conv.weight", (filt, d, k), TRAIN, conv=True), Entry(c + "conv1d_1.conv.bias", (filt,), TRAIN),
        Entry(c + "layer_norm_1.weight", (filt,), TRAIN), Entry(c + "layer_norm_1.bias", (filt,), TRAIN),
        Entry(c + "conv1d_2.conv.weight", (filt, filt, k), TRAIN, conv=True), Entry(c + "conv1d_2.conv.bias", (filt,), TRAIN),
        Entry(c + "layer_norm_2.weight", (filt,), TRAIN), Entry(c + "layer_norm_2.bias", (filt,), TRAIN),
        Entry(pre + "linear_layer.weight", (1, filt), TRAIN), Entry(pre + "linear_layer.bias", (1,), TRAIN),
    ]


def _cnn_scalar(pre, size_one, size_two, reduce=30):
    """CWT pitch mean/std heads (reference: model/modules.py:358-385): parameters exist, compute is off."""
    out = []
    for name, size in (("flat_one", size_one), ("flat_two", size_two)):
        out += [Entry(pre + name + ".net.0.weight", (1, size, 1), UNUSED), Entry(pre + name + ".net.0.bias", (1,), UNUSED),
                Entry(pre + name + ".net.2.weight", (reduce,), UNUSED), Entry(pre + name + ".net.2.bias", (reduce,), UNUSED)]
    out += [Entry(pre + "linear.weight", (1, reduce), UNUSED), Entry(pre + "linear.bias", (1,), UNUSED)]
    return out


def build_entries(model_config, n_mel, n_speakers, n_vocab):
    """Entries in FORWARD order (the flat buffer follows it, so backward completes buckets from the end).
    reference shapes: fs_two/transformer/{Models,Layers,SubLayers}.py, fs_two/model/{fastspeech2,modules}.py."""
    tr = model_config["transformer"]
    d = tr["encoder_hidden"]
    assert tr["decoder_hidden"] == d and tr["variance_hidden"] == d, "kernels assume one hidden size"
    d_ff = tr["conv_filter_size"]
    k1, k2 = tr["conv_kernel_size"]
    n_pos = model_config["max_seq_len"] + 1
    vp = model_config["variance_predictor"]
    n_bins = model_config["variance_embedding"]["n_bins"]
    e = [Entry("encoder.position_enc", (1, n_pos, d), FROZEN), Entry("encoder.src_word_emb.weight", (n_vocab, d), TRAIN)]
    for i in range(tr["encoder_layer"]):
        e += _fft_block("encoder.layer_stack.%d." % i, d, d_ff, k1, k2)
    e.append(Entry("speaker_emb.weight", (n_speakers, d), TRAIN))
    va = "variance_adaptor."
    for name in ("duration_predictor", "pitch_predictor", "energy_predictor"):
        e += _predictor(va + name + ".", d, vp["filter_size"], vp["kernel_size"])
    e += _cnn_scalar(va + "pitch_mean.", d, 11) + _cnn_scalar(va + "pitch_std.", d, 11)
    e += [Entry(va + "pitch_bins", (n_bins - 1,), FROZEN), Entry(va + "energy_bins", (n_bins - 1,), FROZEN),
          Entry(va + "pitch_embedding.weight", (n_bins, d), TRAIN), Entry(va + "energy_embedding.weight", (n_bins, d), TRAIN),
          Entry("decoder.position_enc", (1, n_pos, d), FROZEN)]
    for i in range(tr["decoder_layer"]):
        e += _fft_block("decoder.layer_stack.%d." % i, d, d_ff, k1, k2)
    e += [Entry("mel_linear.weight", (n_mel, d), TRAIN), Entry("mel_linear.bias", (n_mel,), TRAIN)]
    chans = [n_mel, 512, 512, 512, 512, n_mel]                       # PostNet(): Layers.py:76-82
    for i in range(5):
        p = "postnet.convolutions.%d." % i
        ci, co = chans[i], chans[i + 1]
        e += [Entry(p + "0.conv.weight", (co, ci, 5), TRAIN, conv=True), Entry(p + "0.conv.bias", (co,), TRAIN),
              Entry(p + "1.weight", (co,), TRAIN), Entry(p + "1.bias", (co,), TRAIN),
              Entry(p + "1.running_mean", (co,), BUFFER), Entry(p + "1.running_var", (co,), BUFFER),
              Entry(p + "1.num_batches_tracked", (), BUFFER, dtype="int64")]
    return e


def layout(entries, align=8):
    """Assign flat offsets to the TRAIN entries (each aligned to `align` elements = 16 bytes of bf16).
    Returns (OrderedDict key -> Entry, total padded length)."""
    off = 0
    table = OrderedDict()
    for en in entries:
        if en.kind == TRAIN:
            en.offset = off
            off += (en.numel + align - 1) // align * align
        table[en.key] = en
    return table, off


def buckets(table, total, bucket_elems):
    """Contiguous [start, end) ranges of the flat gradient buffer, at most ~bucket_elems long, cut at parameter
    boundaries, listed from the END of the buffer (the order in which backward finishes them)."""
    cuts = sorted({en.offset for en in table.values() if en.kind == TRAIN} | {total})
    out, end = [], total
    start_candidates = cuts[:-1]
    i = len(start_candidates) - 1
    while end > 0:
        j = i
        while j > 0 and end - start_candidates[j - 1] <= bucket_elems:
            j -= 1
        start = start_candidates[j]
        out.append((start, end))
        end = start
        i = j - 1
    return out


def reference_parameter_keys(entries_table):
    """Keys in the order of the reference model's `model.parameters()` — the indices torch.optim.Adam's `state_dict()`
    uses (reference: fs_two/model/optimizer.py:10-15 builds Adam over `model.parameters()`, train.py:221 saves its
    `state_dict()`).  Order there (a module's own Parameters, then its children in registration order): encoder (position_enc, src_word_emb, layer_stack) ->
    variance_adaptor (own Parameters pitch_bins / energy_bins first, then predictors, pitch_mean / pitch_std,
    embeddings: model/modules.py:20-90) -> decoder -> mel_linear -> speaker_emb -> postnet (model/fastspeech2.py:21-41);
    inside a block w_qs, w_ks, w_vs, layer_norm, fc (SubLayers.py:14-29), each weight before its bias.
    BatchNorm running statistics are buffers, not parameters."""
    keys = [k for k, en in entries_table.items() if en.kind != BUFFER]

    def rank(k):
        parts = k.split(".")
        top = {"encoder": 0, "variance_adaptor": 1, "decoder": 2, "mel_linear": 3, "speaker_emb": 4, "postnet": 5}[parts[0]]
        r = [top]
        if parts[0] in ("encoder", "decoder"):
            if parts[1] == "position_enc":      # a Parameter of Encoder/Decoder itself: before the children's
                r += [0]
            elif parts[1] == "src_word_emb":
                r += [1]
            else:
                sub = parts[3] + "." + parts[4]
                order = ["slf_attn.w_qs", "slf_attn.w_ks", "slf_attn.w_vs", "slf_attn.layer_norm", "slf_attn.fc", "pos_ffn.w_1",
                         "pos_ffn.w_2", "pos_ffn.layer_norm"]
                r += [2, int(parts[2]), order.index(sub), 0 if parts[-1] == "weight" else 1]
        elif parts[0] == "variance_adaptor":
            names = ["pitch_bins", "energy_bins", "duration_predictor", "pitch_predictor", "energy_predictor", "pitch_mean", "pitch_std",
                     "pitch_embedding", "energy_embedding"]
            r += [names.index(parts[1])]
            if parts[1].endswith("_predictor"):
                sub = ".".join(parts[2:-1])
                order = ["conv_layer.conv1d_1.conv", "conv_layer.layer_norm_1", "conv_layer.conv1d_2.conv", "conv_layer.layer_norm_2", "linear_layer"]
                r += [order.index(sub), 0 if parts[-1] == "weight" else 1]
            elif parts[1] in ("pitch_mean", "pitch_std"):
                sub = ".".join(parts[2:-1])
                order = ["flat_one.net.0", "flat_one.net.2", "flat_two.net.0", "flat_two.net.2", "linear"]
                r += [order.index(sub), 0 if parts[-1] == "weight" else 1]
        elif parts[0] == "postnet":
            r += [int(parts[2]), int(parts[3]), 0 if parts[-1] == "weight" else 1]
        else:
            r += [0 if parts[-1] == "weight" else 1]
        return r
    return sorted(keys, key=rank)
