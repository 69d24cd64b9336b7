"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the reference's HiFi-GAN V1
generator inference, as plain functions over the generator `state_dict` (reference key names).
See oracle/fs2.py for the import rule and for how the oracle is pinned (tests/golden/hifi_*.npz)."""
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # reference: hifi/models.py:9


def fold_weight_norm(sd):
    """reference: hifi/models.py:203-210 (torch.nn.utils.remove_weight_norm, dim=0):
    w = g * v / ||v|| with the norm over every dim except 0 — for ConvTranspose1d dim 0 is IN-channels.
    Accepts a weight-normed state_dict (`*.weight_g`/`*.weight_v`) and returns a folded one (`*.weight`)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len("weight_g")]
            wv = sd[base + "weight_v"]
            norm = wv.reshape(wv.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (wv.dim() - 1)))
            out[base + "weight"] = wv * (v / norm)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v
    return out


def res_block1(sd, pre, x, k, dilations):
    """reference: hifi/models.py:88-95 — three (dilated conv, conv) pairs with residuals."""
    for j, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, sd[pre + "convs1.%d.weight" % j], sd[pre + "convs1.%d.bias" % j],
                      dilation=d, padding=(k * d - d) // 2)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, sd[pre + "convs2.%d.weight" % j], sd[pre + "convs2.%d.bias" % j],
                      padding=(k - 1) // 2)
        x = xt + x
    return x


def generator(sd, h, mel):
    """reference: hifi/models.py:185-201.  `sd` must be folded (no weight_g/weight_v).
    mel (B,80,T) -> (B,1,T*prod(upsample_rates)); last LeakyReLU uses the default slope 0.01."""
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, sd["ups.%d.weight" % i], sd["ups.%d.bias" % i], stride=u,
                               padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            y = res_block1(sd, "resblocks.%d." % (i * nk + j), x, rk, rd)
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def to_int16(audio, max_wav_value=32768):
    """reference: hifiapi.py:49-51 — scale, then numpy astype('int16') (C truncation toward zero)."""
    return (audio * max_wav_value).cpu().numpy().astype("int16")
