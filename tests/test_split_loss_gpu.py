"""The loss in two halves on two streams (ops.fs2_loss_split / fs2_loss_finalize; include/ttsk.h: ttsk_fs2_loss_mel / _var / _finalize): the same
values, bit for bit, as the one-launch ttsk_fs2_loss (reference: fs_two/model/loss.py:24-134), and a training step that takes it that way —
the frame-level half on the step's stream without waiting for the predictors, the rest on the predictors' stream — ends where the one-launch
step ends."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def cfg():
    from tts_king_amd.config import default_config
    return default_config()


@pytest.mark.parametrize("B,T,L,limit", [(16, 423, 64, None), (3, 57, 11, None), (5, 120, 30, 97), (1, 1, 1, None)])
def test_two_stream_loss_is_the_one_launch_loss_bit_for_bit(B, T, L, limit):
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    nm = 80
    mel, post, mel_t = (torch.randn(B, T, nm, generator=g).to(DEV) for _ in range(3))
    mel_lens = torch.randint(1, T + 1, (B,), generator=g).to(DEV)
    src_lens = torch.randint(1, L + 1, (B,), generator=g).to(DEV)
    pitch, energy, logd, pitch_t, energy_t = (torch.randn(B, L, generator=g).to(DEV) for _ in range(5))
    dur_t = torch.randint(0, 9, (B, L), generator=g).to(DEV)
    fl = None if limit is None else (torch.tensor([limit], dtype=torch.int32, device=DEV), 0)
    ref = ops.fs2_loss(mel, post, mel_t, mel_lens, pitch, energy, logd, pitch_t, energy_t, dur_t, src_lens, grad_scale=0.25, frame_limit=fl)
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream())
    *got, pending = ops.fs2_loss_split(mel, post, mel_t, mel_lens, pitch, energy, logd, pitch_t, energy_t, dur_t, src_lens, side, grad_scale=0.25,
                                        frame_limit=fl)
    torch.cuda.current_stream().wait_stream(side)
    ops.fs2_loss_finalize(got[0], pending, src_lens)
    torch.cuda.synchronize()
    for name, a, b in zip(("losses", "dmel_sum", "dpost", "dpitch", "denergy", "dlogd"), ref, got):
        assert torch.equal(a, b), name
    assert float(ref[0][0]) > 0.0


@pytest.mark.parametrize("graphed", [False, True])
def test_training_with_the_two_stream_loss_ends_where_the_one_launch_loss_ends(cfg, graphed):
    """Three optimizer steps, dropout on, ragged lengths; `split_loss` on and off; plain launches and a replayed hipGraph: the same losses
    and the same parameters, bit for bit (the kernels and their inputs are the same; only the streams they are queued on differ)."""
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.graph import GraphedTrainStep, make_enqueue
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.synthetic import make_batch
    from tts_king_amd.train_step import to_device
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 1
    batch = to_device(make_batch(6, 40, seed=11, ragged=True), DEV)
    res = []
    for split in (False, True):
        m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV, seed=5).train()
        m.split_loss = split
        opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
        enq = make_enqueue(m, opt, c, FastSpeech2Loss(c.preprocess_config, c.model_config))
        ls = []
        if graphed:
            gr = GraphedTrainStep(enq, batch, warmup=0)
            for _ in range(3):
                ls.append(gr.run()[0].cpu().clone())
        else:
            for _ in range(3):
                ls.append(enq(batch)[0].cpu().clone())
        torch.cuda.synchronize()
        assert m._var_on_pred is False and m._loss_finalize is None and not m._pred_fwd_pending
        res.append((ls, m.flat_buffers()[0].cpu().clone()))
    (la, pa), (lb, pb) = res
    for a, b in zip(la, lb):
        assert torch.equal(a, b)
    assert torch.equal(pa, pb)
    assert float(la[0][0]) != float(la[2][0])          # (the steps did train)


@pytest.mark.parametrize("pred_side", ["f", "b", "0"])
def test_every_predictor_stream_setting_takes_the_same_steps(cfg, pred_side):
    """TTSK_PRED_SIDE = f (the predictors' forward alone on their stream: the two-stream loss is taken and the predictors' backward, on the
    step's stream, has to wait for its other half), b (no pending stream at the loss: one launch), 0: the same two steps, bit for bit, as
    the default."""
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.graph import make_enqueue
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.synthetic import make_batch
    from tts_king_amd.train_step import to_device
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 1
    batch = to_device(make_batch(4, 36, seed=13, ragged=True), DEV)
    res = []
    for ps in ("1", pred_side):
        m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV, seed=9).train()
        m.pred_side = ps
        opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
        enq = make_enqueue(m, opt, c, FastSpeech2Loss(c.preprocess_config, c.model_config))
        ls = [enq(batch)[0].cpu().clone() for _ in range(2)]
        torch.cuda.synchronize()
        assert m._var_on_pred is False and m._loss_finalize is None and not m._pred_fwd_pending
        res.append((ls, m.flat_buffers()[0].cpu().clone()))
    (la, pa), (lb, pb) = res
    for a, b in zip(la, lb):
        assert torch.equal(a, b)
    assert torch.equal(pa, pb)
