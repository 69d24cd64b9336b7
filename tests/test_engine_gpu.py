"""GPU: the trainer's step engine (tts_king_amd/engine.py; reference loop: train.py:78-235) — shape-bucketed batches carry the
frame count the reference would have seen, so a bucketed step equals the unpadded step (checked against the oracle on the
UNPADDED batch), and a loop of replayed hipGraphs ends in bit-identical weights to the same loop launched eagerly."""
import copy
import math

import numpy as np
import pytest
import torch

from oracle import fs2 as ofs2
from tests.oracle_util import fs2_state_dict, rel_rms
from tests.test_parity_gpu import build, no_dropout_config, oracle_without_dropout
from tts_king_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def as_numpy(b):
    return tuple(x.numpy() if torch.is_tensor(x) else x for x in b)


def padded_device_batch(b, l_bucket=8, t_bucket=32):
    from tts_king_amd.engine import PaddedBatch, pad_to_bucket
    from tts_king_amd.train_step import to_device
    p = pad_to_bucket(as_numpy(b), l_bucket, t_bucket, 1000)
    d = PaddedBatch(to_device(p, DEV))
    d.t_true, d.l_true = p.t_true, p.l_true
    d.frame_limit = torch.tensor([p.t_true], dtype=torch.int32, device=DEV) if p.t_true is not None else None
    d.phoneme_limit = torch.full((len(p[0]),), p.l_true, dtype=torch.int64, device=DEV) if p.l_true is not None else None
    return d


def test_bucketed_step_equals_the_unpadded_step(cfg):
    """B=3 ragged, L=37 -> 40, T=217 -> 224: forward, loss, backward on the padded batch with `frame_limit` against the
    oracle on the unpadded batch (losses 1 %, global gradient norm 2 %), BatchNorm running statistics included."""
    from tts_king_amd import ops
    b = make_batch(3, 37, seed=91, ragged=True)
    T_true = int(b[8])
    pb = padded_device_batch(b)
    assert int(pb[5]) == 40 and int(pb[8]) % 32 == 0 and int(pb[8]) > T_true and pb.t_true == T_true and pb.l_true == 37
    m = build(cfg, 7, dropout=False).train()
    with torch.no_grad():
        out, ctx = m._forward(True, pb[2], pb[3], pb[4], int(pb[5]), pb[7], pb[8], pb[9], pb[10], pb[11], 1.0, 1.0, 1.0, frame_limit=pb.frame_limit,
                              phoneme_limit=pb.phoneme_limit)
        losses, dmel_sum, dpost, dp, de, dd = ops.fs2_loss(out[0], out[8], pb[6], pb[7], out[1], out[2], out[3], pb[11], pb[9], pb[10], pb[4],
                                                           grad_scale=1.0, frame_limit=(pb.frame_limit, 0))
        m.backward_native(ctx, dmel_sum, dpost, dp, de, dd)
    torch.cuda.synchronize()
    tr = ofs2.OracleTrainer(fs2_state_dict(cfg, 7), no_dropout_config(cfg), cfg.train_config, 0)
    bn = {}
    with oracle_without_dropout():
        o = ofs2.fs2_forward(tr.sd, tr.mc, *b[2:], train=True, bn_buffers=bn)
        ls = ofs2.fs2_loss(b, o)
        ls[0].sum().backward()
    got, want = losses.cpu().tolist(), [float(l.sum()) for l in ls]
    print("bucketed losses", [round(v, 5) for v in got[:5]], "oracle (unpadded)", [round(v, 5) for v in want[:5]])
    np.testing.assert_allclose(got[:5], want[:5], rtol=0.01)
    r = rel_rms(out[0][:, :T_true].float().cpu(), o[0].detach())
    assert r <= 0.01, r
    named = dict(m.named_parameters())
    gn = math.sqrt(sum(float(named[k].grad.double().pow(2).sum()) for k in tr.keys))
    print("bucketed global grad norm %.5f oracle %.5f" % (gn, tr.grad_norm()))
    assert abs(gn - tr.grad_norm()) <= 0.02 * tr.grad_norm()
    for grp in ("postnet.convolutions.0", "postnet.convolutions.4", "mel_linear", "decoder.layer_stack.5", "encoder.layer_stack.0"):
        a = math.sqrt(sum(float(named[k].grad.double().pow(2).sum()) for k in tr.keys if k.startswith(grp + ".")))
        w = math.sqrt(sum(float(tr.sd[k].grad.double().pow(2).sum()) for k in tr.keys if k.startswith(grp + ".")))
        assert abs(a - w) <= 0.06 * w, (grp, a, w)
    for k, v in bn.items():                       # running statistics: counted over B * T_true rows, as the reference does
        np.testing.assert_allclose(m.state_dict()[k].cpu().numpy(), v.numpy(), rtol=2e-2, atol=2e-3)
    # and WITHOUT the frame limit the padded batch gives different mel losses (the denominators grow): the limit matters
    m2 = build(cfg, 7, dropout=False).train()
    with torch.no_grad():
        out2, _ = m2._forward(True, pb[2], pb[3], pb[4], int(pb[5]), pb[7], pb[8], pb[9], pb[10], pb[11], 1.0, 1.0, 1.0)
        l2, *_ = ops.fs2_loss(out2[0], out2[8], pb[6], pb[7], out2[1], out2[2], out2[3], pb[11], pb[9], pb[10], pb[4], grad_scale=1.0)
    assert abs(float(l2[1]) - want[1]) > 0.02 * want[1]


def test_graphed_loop_ends_in_the_eager_loops_weights(cfg):
    """16 steps over 2 shape buckets, dropout ON, grad_acc_step 2 (both graph variants): hip_graph on/off -> bit-identical
    parameters, Adam moments and losses; the graphed run captured and replayed."""
    from tts_king_amd.engine import TrainEngine
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 2
    # two shape buckets x (accumulate, update): every key is seen four times -> eager, captured, replayed, replayed
    shapes = [(2, (30, 29)[i % 2] if (i // 2) % 2 == 0 else (44, 43)[i % 2], i + 1) for i in range(16)]
    finals = []
    for graphed in (True, False):
        m = build(c, 7, dropout=True)
        opt = ScheduledOptim(m, c.train_config, c.model_config, 50)
        eng = TrainEngine(m, opt, c, FastSpeech2Loss(c.preprocess_config, c.model_config), hip_graph=graphed)
        seen = []
        for step, (B, L, seed) in enumerate(shapes, 1):
            b = make_batch(B, L, seed=400 + seed, ragged=True, dur_hi=6)
            losses, _ = eng.step(padded_device_batch(b, 8, 64), step)
            seen.append(losses.cpu().tolist())
        torch.cuda.synchronize()
        finals.append((m.flat_buffers()[0].clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), seen, dict(eng.stats), opt.current_step, opt._host_step))
    g, e = finals
    print("engine stats graphed", g[4], "eager", e[4])
    assert g[4]["captured"] >= 3 and g[4]["replayed"] >= 2 and e[4]["captured"] == 0
    assert g[5] == e[5] == 58 and g[6] == e[6]
    assert g[3] == e[3], "losses differ between the graphed and the eager loop"
    assert torch.equal(g[0], e[0]) and torch.equal(g[1], e[1]) and torch.equal(g[2], e[2])
