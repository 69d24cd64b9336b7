"""GPU: the data-parallel step path on ONE rank over RCCL (world_size 1): bucketed all-reduce issued from backward's
group notifications, deferred split-K reducers flushed before each bucket, clip + Adam after `finish()` — the code path
`bench.py --gpus N` runs, minus the peers.  Gradients and the weight update must equal the single-GPU step's."""
import copy
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_reducer_step_equals_plain_step(cfg):
    from tests.oracle_util import fs2_state_dict
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.graph import make_enqueue
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.parallel import GradReducer
    from tts_king_amd.synthetic import make_batch
    from tts_king_amd.train_step import to_device
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        created = True
    try:
        c = copy.deepcopy(cfg)
        c.train_config["optimizer"]["grad_acc_step"] = 1
        batch = to_device(make_batch(4, 32, seed=8, ragged=True), DEV)
        res = []
        for use_reducer in (False, True):
            m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV)
            m.load_state_dict(fs2_state_dict(c, 7))
            m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0
            m.train()
            opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
            red = GradReducer(m.flat_buffers()[1], m.grad_buckets(8), m.group_offsets()) if use_reducer else None
            enq = make_enqueue(m, opt, c, FastSpeech2Loss(c.preprocess_config, c.model_config), reducer=red,
                               grad_scale=red.grad_scale(1) if red else None)
            losses, _ = enq(batch)
            torch.cuda.synchronize()
            if red is not None:
                assert len(red.launched) == len(red.buckets) and red.launched[0][1] == m.flat_buffers()[1].numel()
            res.append((losses.cpu().clone(), m.flat_buffers()[0].cpu().clone()))
        assert torch.equal(res[0][0], res[1][0])
        assert torch.equal(res[0][1], res[1][1])          # identical weights after clip + Adam: same gradients
    finally:
        if created:
            dist.destroy_process_group()


def test_reducer_step_is_graph_capturable(cfg):
    """The RCCL all-reduces are captured into the step's hipGraph (bench.py replays the data-parallel step too): three
    replays leave exactly the weights of three eager steps."""
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.graph import GraphedTrainStep, make_enqueue
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.parallel import GradReducer
    from tts_king_amd.synthetic import make_batch
    from tts_king_amd.train_step import to_device
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = False
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        created = True
    try:
        c = copy.deepcopy(cfg)
        c.train_config["optimizer"]["grad_acc_step"] = 1
        batch = to_device(make_batch(4, 32, seed=8, ragged=True), DEV)
        outs = []
        for graphed in (False, True):
            m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV, seed=5)
            m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0
            m.train()
            opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
            red = GradReducer(m.flat_buffers()[1], m.grad_buckets(8), m.group_offsets())
            enq = make_enqueue(m, opt, c, FastSpeech2Loss(c.preprocess_config, c.model_config), reducer=red, grad_scale=red.grad_scale(1))
            if graphed:
                g = GraphedTrainStep(enq, batch, warmup=0)          # capturing does not execute the step
                for _ in range(3):
                    g.run()
            else:
                for _ in range(3):
                    enq(batch)
            torch.cuda.synchronize()
            outs.append((opt.current_step, m.flat_buffers()[0].cpu().clone()))
        assert outs[0][0] == outs[1][0] == 3
        assert torch.equal(outs[0][1], outs[1][1])
    finally:
        if created:
            dist.destroy_process_group()
