"""CPU: the oracle (oracle/) reproduces the outputs of the reference recorded in tests/golden/.

Tolerance: fp32 vs fp32 on the same torch build; the restatement uses the same ATen ops in a different
association order in a few places, so rtol 1e-4 / atol 2e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import fs2 as ofs2
from oracle import hifigan as ohifi
from tests.oracle_util import fs2_state_dict, hifi_state_dict_wn, GOLDEN
from tts_king_amd.synthetic import make_batch, make_mel

RT, AT = 1e-4, 2e-5


def close(a, b, rtol=RT, atol=AT):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def test_eval_teacher_forced(cfg):
    g = np.load(os.path.join(GOLDEN, "fs2_eval_tf.npz"))
    sd = fs2_state_dict(cfg, int(g["weight_seed"]))
    b = make_batch(int(g["B"]), int(g["L"]), seed=int(g["seed"]), ragged=True)
    with torch.no_grad():
        o = ofs2.fs2_forward(sd, cfg.model_config, *b[2:], train=False)
    close(o[0], g["mel"]); close(o[9], g["post"]); close(o[1], g["pitch"]); close(o[2], g["energy"])
    close(o[3], g["logd"]); assert o[8].tolist() == g["mel_lens"].tolist()


def test_eval_free_running(cfg):
    g = np.load(os.path.join(GOLDEN, "fs2_eval_free.npz"))
    sd = fs2_state_dict(cfg, int(g["weight_seed"]))
    sd["variance_adaptor.duration_predictor.linear_layer.bias"].fill_(float(g["dur_bias"]))
    b = make_batch(int(g["B"]), int(g["L"]), seed=int(g["seed"]), ragged=True)
    dc, pc, ec = [float(x) for x in g["controls"]]
    with torch.no_grad():
        o = ofs2.fs2_forward(sd, cfg.model_config, b[2], b[3], b[4], b[5], d_control=dc, p_control=pc,
                             e_control=ec, train=False)
    assert o[4].dtype == torch.float32
    np.testing.assert_array_equal(o[4].numpy(), g["d_rounded"])
    assert o[8].tolist() == g["mel_lens"].tolist()
    close(o[0], g["mel"]); close(o[9], g["post"]); close(o[1], g["pitch"]); close(o[2], g["energy"])


def test_train_step_no_dropout(cfg):
    g = np.load(os.path.join(GOLDEN, "fs2_train_p0.npz"))
    sd = fs2_state_dict(cfg, int(g["weight_seed"]))
    keys = ofs2.trainable_keys(sd)
    assert sorted(keys) == sorted(str(k) for k in g["grad_keys"])
    for k in keys:
        sd[k].requires_grad_(True)
    b = make_batch(int(g["B"]), int(g["L"]), seed=int(g["seed"]), ragged=True)
    mc = cfg.model_config
    import copy
    mc0 = copy.deepcopy(mc)
    mc0["transformer"]["encoder_dropout"] = 0.0
    mc0["transformer"]["decoder_dropout"] = 0.0
    mc0["variance_predictor"]["dropout"] = 0.0
    # postnet dropout is hard-coded (0.5) in the reference; the golden was made with dropout disabled
    orig = ofs2._drop
    ofs2._drop = lambda x, p, train: x
    try:
        bufs = {}
        o = ofs2.fs2_forward(sd, mc0, *b[2:], train=True, bn_buffers=bufs)
        ls = ofs2.fs2_loss(b, o)
        ls[0].sum().backward()
    finally:
        ofs2._drop = orig
    assert ls[0].shape == (1,)
    close([float(l.sum()) for l in ls], g["losses"], rtol=1e-5)
    close(o[0].detach(), g["mel"]); close(o[9].detach(), g["post"], atol=1e-4)
    gn = {str(k): float(v) for k, v in zip(g["grad_keys"], g["grad_norms"])}
    for k in keys:
        assert abs(float(sd[k].grad.norm()) - gn[k]) <= 2e-4 * gn[k] + 1e-6, k
    for name in g.files:
        if name.startswith("grad/"):
            close(sd[name[5:]].grad, g[name], rtol=2e-4, atol=1e-5)
        if name.startswith("bn/"):
            close(bufs[name[3:]], g[name], rtol=1e-5, atol=1e-6)


def test_length_regulator_edges():
    g = np.load(os.path.join(GOLDEN, "length_regulator.npz"))
    x, d = torch.from_numpy(g["x"]), torch.from_numpy(g["d"])
    for tag, ml in (("none", None), ("crop4", 4), ("pad12", 12)):
        o, n = ofs2.length_regulator(x, d, ml)
        np.testing.assert_array_equal(o.numpy(), g["out_" + tag])
        np.testing.assert_array_equal(n.numpy(), g["len_" + tag])
    xi, di = torch.from_numpy(g["xi"]), torch.from_numpy(g["di"])
    o, n = ofs2.length_regulator(xi, di, int(di.sum(1).max()))
    np.testing.assert_array_equal(o.numpy(), g["out_int"])
    np.testing.assert_array_equal(n.numpy(), g["len_int"])


@pytest.mark.parametrize("s", [1, 4000, 300001])
def test_adam_clip_lr_step(cfg, s):
    g = np.load(os.path.join(GOLDEN, "adam_steps.npz"))
    sd = fs2_state_dict(cfg, int(g["weight_seed"]))
    b = make_batch(int(g["B"]), int(g["L"]), seed=int(g["seed"]), ragged=True)
    import copy
    tc = copy.deepcopy(cfg.train_config)
    tc["optimizer"]["grad_acc_step"] = 1
    tr = ofs2.OracleTrainer(sd, cfg.model_config, tc, current_step=s - 1)
    before = {k: v.detach().clone() for k, v in tr.sd.items()}
    orig = ofs2._drop
    ofs2._drop = lambda x, p, train: x
    try:
        out = ofs2.fs2_forward(tr.sd, tr.mc, *b[2:], train=True, bn_buffers={})
        ofs2.fs2_loss(b, out)[0].sum().backward()
    finally:
        ofs2._drop = orig
    assert abs(tr.grad_norm() - float(g["gnorm_%d" % s])) < 1e-3 * float(g["gnorm_%d" % s])
    tr.optimizer_step()
    assert abs(ofs2.lr_at(s) - float(g["lr_%d" % s])) < 1e-12
    for name in g.files:
        if name.startswith("delta_%d/" % s):
            k = name.split("/", 1)[1]
            close((tr.sd[k] - before[k]).detach(), g[name], rtol=2e-3, atol=2.5e-7)  # fp32 ulp of the params


def test_hifigan(cfg):
    g = np.load(os.path.join(GOLDEN, "hifi_b2_t32.npz"))
    sdw = hifi_state_dict_wn(int(g["weight_seed"]))
    assert len(sdw) == int(g["n_wn_keys"])
    sd = ohifi.fold_weight_norm(sdw)
    assert len(sd) == int(g["n_folded_keys"])
    for name in g.files:
        if name.startswith("fold/"):
            close(sd[name[5:]].reshape(-1)[:64], g[name], rtol=1e-5, atol=1e-7)
    mel = make_mel(int(g["B"]), int(g["T"]), seed=int(g["seed"]))
    with torch.no_grad():
        wav = ohifi.generator(sd, cfg.hifi, mel)
    close(wav, g["wav"], rtol=1e-4, atol=2e-6)
    i16 = ohifi.to_int16(wav)
    assert np.abs(i16.astype(np.int32) - g["int16"].astype(np.int32)).max() <= 1
