"""GPU: the data-parallel step path on ONE rank over RCCL (world_size 1): bucketed all-reduce issued from backward's
group notifications, deferred split-K reducers flushed before each bucket, clip + Adam after `finish()` — the code path
`bench.py --gpus N` runs, minus the peers.  Gradients and the weight update must equal the single-GPU step's."""
import copy
import os

import pytest
import torch
import torch.distributed as dist

from tests.conftest import isolated

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("schedule", ["side", "late", "per_bucket"])
def test_reducer_step_equals_plain_step(cfg, schedule):
    """Every data-parallel schedule (FastSpeech2.dp_schedule) leaves the plain step's losses and weights, bit for bit, and launches
    every bucket exactly once, from the end of the buffer."""
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
            if schedule == "per_bucket":        # no second stream: buckets are flushed one by one on the main stream as they complete
                m.dw_side_wgs = 0               # (the grouped GEMMs carry every weight gradient there: compare like with like)
                m.dwconv = False
            else:
                m.dp_schedule = schedule
            m.train()
            opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
            red = GradReducer(m.flat_buffers()[1], m.grad_buckets(8), m.group_offsets(), force_collectives=True) if use_reducer else None
            enq = make_enqueue(m, opt, c, FastSpeech2Loss(c.preprocess_config, c.model_config), reducer=red,
                               grad_scale=red.grad_scale(1) if red else None)
            losses, _ = enq(batch)
            torch.cuda.synchronize()
            if red is not None:
                assert red.launched == list(red.buckets) and red.launched[0][1] == m.flat_buffers()[1].numel()
            res.append((losses.cpu().clone(), m.flat_buffers()[0].cpu().clone()))
        assert torch.equal(res[0][0], res[1][0])
        assert torch.equal(res[0][1], res[1][1])          # identical weights after clip + Adam: same gradients
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("schedule", ["side", "late"])
def test_buckets_are_final_when_announced_full_size(cfg, schedule):
    """at the training shape (B=16, L=64, T=423) the 80-channel weight gradients (mel_linear, the PostNet's first and
    last conv) are split-K grouped problems; the "side" schedule used to reduce their slabs and announce their buckets BEFORE their
    GEMMs had run.  At world size 1 an in-place all-reduce is the identity, so the final buffer cannot show that — this test looks at
    what a collective would READ: the gradient buffer is poisoned, every bucket is snapshotted at the moment the reducer issues it
    (after draining the issuing stream, eager launches), and each snapshot must already equal the bucket's final content, which in
    turn must equal the plain (no reducer) backward bit for bit."""
    from tests.test_parity_gpu import build
    from tts_king_amd import ops
    from tts_king_amd.parallel import GradReducer
    from tts_king_amd.synthetic import make_batch
    c = copy.deepcopy(cfg)
    b = make_batch(16, 64, seed=1234)
    dev_b = [t.to(DEV) if torch.is_tensor(t) else t for t in b]

    def backward(m, on_bucket=None):
        with torch.no_grad():
            out, ctx = m._forward(True, dev_b[2], dev_b[3], dev_b[4], int(b[5]), dev_b[7], b[8], dev_b[9], dev_b[10], dev_b[11], 1.0, 1.0, 1.0)
            _, dmel_sum, dpost, dp, de, dd = ops.fs2_loss(out[0], out[8], dev_b[6], dev_b[7], out[1], out[2], out[3], dev_b[11],
                                                          dev_b[9], dev_b[10], dev_b[4], grad_scale=1.0)
            m.flat_buffers()[1].fill_(1e30)                       # poison: the overwriting backward must leave none of it
            m.backward_native(ctx, dmel_sum, dpost, dp, de, dd, on_bucket=on_bucket, accumulate=False)
        torch.cuda.synchronize()
        return m.flat_buffers()[1].clone()

    m = build(c, 7, dropout=False).train()
    m.dp_schedule = schedule
    from tts_king_amd import params as P
    owned = torch.zeros(m.flat_buffers()[1].numel(), dtype=torch.bool, device=DEV)      # (the alignment padding between parameters is never written)
    for en in m._table.values():
        if en.kind == P.TRAIN:
            owned[en.offset:en.offset + en.numel] = True
    plain = backward(m)
    assert float(plain[owned].abs().max()) < 1e20
    red = GradReducer(m.flat_buffers()[1], m.grad_buckets(24), m.group_offsets())
    snaps = []

    def on_group_done(name):
        n0 = len(red.launched)
        red.on_group_done(name)
        if len(red.launched) > n0:
            torch.cuda.current_stream().synchronize()            # the stream the collective would be issued from
            for s, e in red.launched[n0:]:
                snaps.append((name, s, e, m.flat_buffers()[1][s:e].clone()))

    m.train()
    final = backward(m, on_bucket=on_group_done)
    red.finish()
    assert red.launched == list(red.buckets)
    assert len(snaps) == len(red.buckets)
    assert torch.equal(final[owned], plain[owned]), "data-parallel %s schedule: gradients differ from the plain backward" % schedule
    for name, s, e, snap in snaps:
        bad = int(((snap != final[s:e]) & owned[s:e]).sum())
        assert bad == 0, "bucket [%d, %d) announced with group %r held %d elements that changed afterwards" % (s, e, name, bad)


@isolated
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


def test_two_virtual_ranks_equal_reference_grad_accumulation(cfg):
    """data parallelism over N ranks = the reference's `grad_acc_step = N` (train.py:43-47).  Two "virtual ranks"
    on this one GPU: micro-batches of seeds 1234 and 1235 at full size (B=16, L=64), each through the HIP forward / backward with
    `grad_scale = 1/2` into a zeroed flat gradient buffer — what each rank holds before the all-reduce; their SUM (what the SUM
    all-reduce leaves on every rank) must equal (a) the oracle's accumulated `grad_acc_step = 2` gradients (global norm 2 %,
    per-group norm 6 %, a few whole tensors rel-RMS 8 %) and (b) what the HIP `grad_acc_step = 2` path accumulates in place.
    (b) is bit-for-bit: every producer adds its finished fp32 value to the buffer once per micro-step (measured: 0 of 34.6 M
    elements differ)."""
    import math
    from oracle import fs2 as ofs2
    from tests.oracle_util import fs2_state_dict, rel_rms
    from tests.test_parity_gpu import GROUPS, build, no_dropout_config, oracle_without_dropout
    from tts_king_amd import ops
    from tts_king_amd.synthetic import make_batch
    c = copy.deepcopy(cfg)
    batches = [make_batch(16, 64, seed=1234), make_batch(16, 64, seed=1235)]
    m = build(c, 7, dropout=False).train()
    flat_grad = m.flat_buffers()[1]

    def micro(b, scale):
        dev_b = [t.to(DEV) if torch.is_tensor(t) else t for t in b]
        with torch.no_grad():
            out, ctx = m._forward(True, dev_b[2], dev_b[3], dev_b[4], int(b[5]), dev_b[7], b[8], dev_b[9], dev_b[10], dev_b[11], 1.0, 1.0, 1.0)
            losses, dmel_sum, dpost, dp, de, dd = ops.fs2_loss(out[0], out[8], dev_b[6], dev_b[7], out[1], out[2], out[3], dev_b[11],
                                                               dev_b[9], dev_b[10], dev_b[4], grad_scale=scale)
            m.backward_native(ctx, dmel_sum, dpost, dp, de, dd)
        torch.cuda.synchronize()
        return losses.cpu()

    # BatchNorm running statistics advance with every training forward; they do not enter the gradients (batch statistics do)
    per_rank = []
    for b in batches:                       # virtual rank r: its own micro-batch into a zeroed buffer
        flat_grad.zero_()
        micro(b, 0.5)
        per_rank.append(flat_grad.clone())
    reduced = per_rank[0] + per_rank[1]     # the SUM all-reduce (fp32, two addends: order-independent)
    flat_grad.zero_()
    for b in batches:                       # the HIP grad_acc_step = 2 path: both micro-steps accumulate in place
        micro(b, 0.5)
    accumulated = flat_grad.clone()
    diff = (reduced - accumulated).abs()
    n_diff = int((diff > 0).sum())
    print("virtual ranks vs in-place accumulation: %d of %d elements differ, max abs %.3e (|g| max %.3e)"
          % (n_diff, diff.numel(), float(diff.max()), float(accumulated.abs().max())))
    assert n_diff == 0, "sum of per-rank gradients != in-place accumulation"      # every producer adds its finished fp32 value once
    # ---- the oracle's grad_acc_step = 2 gradients (before its optimizer step)
    c2 = copy.deepcopy(c)
    c2.train_config["optimizer"]["grad_acc_step"] = 2
    tr = ofs2.OracleTrainer(fs2_state_dict(c, 7), no_dropout_config(c), c2.train_config, 0)
    keep_step = tr.optimizer_step
    tr.optimizer_step = lambda: None        # keep the accumulated .grad to look at
    with oracle_without_dropout():
        tr.train_step(batches[0], 1)
        tr.train_step(batches[1], 2)
    tr.optimizer_step = keep_step
    flat_grad.copy_(reduced)                # read the reduced gradients through the parameters' .grad views
    named = dict(m.named_parameters())
    gsq = osq = 0.0
    worst = (0.0, None)
    for grp in GROUPS:
        a = math.sqrt(sum(float(named[k].grad.double().pow(2).sum()) for k in tr.keys if k.startswith(grp + ".")))
        w = math.sqrt(sum(float(tr.sd[k].grad.double().pow(2).sum()) for k in tr.keys if k.startswith(grp + ".")))
        gsq, osq = gsq + a * a, osq + w * w
        err = abs(a - w) / w
        print("  group %-42s |g| 2 virtual ranks %.5f oracle grad_acc 2 %.5f  (%.2f%%)" % (grp, a, w, 100 * err))
        if err > worst[0]:
            worst = (err, grp)
    for k in ("postnet.convolutions.4.0.conv.weight", "mel_linear.weight", "decoder.layer_stack.5.pos_ffn.w_1.weight",
              "variance_adaptor.energy_embedding.weight", "encoder.layer_stack.3.pos_ffn.w_2.weight", "encoder.src_word_emb.weight"):
        r = rel_rms(named[k].grad.float().cpu(), tr.sd[k].grad)
        print("  grad %-64s rel-RMS vs oracle %.2f%%" % (k, 100 * r))
        assert r <= 0.08, (k, r)
    gn, on = math.sqrt(gsq), math.sqrt(osq)
    print("global grad norm: 2 virtual ranks %.5f, oracle grad_acc_step=2 %.5f; worst group %s" % (gn, on, worst))
    assert abs(gn - on) <= 0.02 * on
    assert worst[0] <= 0.06, worst


@isolated
def test_train_engine_with_reducer_capture_and_fallback(cfg):
    """the trainer's engine (shape buckets, one captured graph per shape) WITH a reducer whose collectives are really
    issued (RCCL communicator of size 1, `force_collectives`): weights after 8 varying-shape steps equal the eager loop's; and a
    capture that fails (forced here) leaves the reducer / step counters where they were and the shape runs eagerly from then on."""
    import numpy as np
    from tts_king_amd import graph as G
    from tts_king_amd.dataset import DeviceFeeder
    from tts_king_amd.engine import TrainEngine
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.parallel import GradReducer
    from tts_king_amd.synthetic import make_batch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29535")
    created = False
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        created = True
    try:
        c = copy.deepcopy(cfg)
        c.train_config["optimizer"]["grad_acc_step"] = 1
        host = [tuple(x.numpy() if torch.is_tensor(x) else x for x in make_batch(4, 30 + (i % 2) * 4, seed=70 + i % 2, ragged=True)) for i in range(8)]
        bucket = (8, 32, int(c.model_config["max_seq_len"]))
        res = {}
        for mode in ("eager", "graph", "capture_fails"):
            m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV, seed=5)
            m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0
            m.train()
            opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
            red = GradReducer(m.flat_buffers()[1], m.grad_buckets(8), m.group_offsets(), force_collectives=True)
            eng = TrainEngine(m, opt, c, FastSpeech2Loss(c.preprocess_config, c.model_config), reducer=red, hip_graph=mode != "eager")
            keep = G.GraphedTrainStep
            if mode == "capture_fails":
                class Boom(keep):
                    def __init__(self, enqueue, example_batch, warmup=2, pool=None, **kw):
                        enqueue(example_batch)          # runs part of a step's Python (and its launches), then fails like a refused capture
                        raise RuntimeError("capture refused (test)")
                import tts_king_amd.engine as E
                E.GraphedTrainStep = Boom
            try:
                step = 0
                for b in DeviceFeeder(host, DEV, bucket=bucket):
                    step += 1
                    eng.step(b, step)
                torch.cuda.synchronize()
            finally:
                if mode == "capture_fails":
                    E.GraphedTrainStep = keep
            res[mode] = (opt.current_step, opt._host_step, m.flat_buffers()[0].cpu().clone(), dict(eng.stats), red._next, len(red._handles))
        assert res["eager"][0] == res["graph"][0] == 8 and res["graph"][1] == 8
        assert res["graph"][3]["captured"] >= 2 and res["graph"][3]["replayed"] >= 4, res["graph"][3]
        assert torch.equal(res["eager"][2], res["graph"][2]), "graphed DP loop != eager DP loop"
        # the forced failure ran one extra (real) step inside the failed "capture" per shape, so the weights differ by design; what
        # must hold: the engine went on eagerly, counted the failure, and the reducer / host step were left consistent
        st = res["capture_fails"][3]
        assert st.get("capture_failed", 0) == 2 and st["captured"] == 0, st
        assert res["capture_fails"][4] == 0 and res["capture_fails"][5] == 0
        assert res["capture_fails"][1] == res["capture_fails"][0] - 2      # host step restored after each failed capture's extra device step
    finally:
        if created:
            dist.destroy_process_group()


@isolated
def test_collective_sequence_is_the_same_replayed_refused_or_eager(cfg):
    """ranks deadlock when they issue different sequences of collectives, and each rank decides by itself, shape by
    shape, whether it replays a graph, captures one, or — when a capture is refused — launches eagerly.  Here three engines over a REAL
    RCCL process group (size 1, collectives forced) run the same nine batches: plain launches; replayed graphs; replayed graphs with the
    first capture refused mid-run (a capture that recorded a whole step and was then thrown away, as RCCL refusing it would leave
    things).  What the reducer issued (or a replay executed) per update step must be identical across the three — every bucket once, in
    bucket order — and so must the weights."""
    from tts_king_amd import graph as G
    from tts_king_amd.dataset import DeviceFeeder
    from tts_king_amd.engine import TrainEngine
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.parallel import GradReducer
    from tts_king_amd.synthetic import make_batch
    import tts_king_amd.engine as E
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29537")
    created = False
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        created = True
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        c = copy.deepcopy(cfg)
        c.train_config["optimizer"]["grad_acc_step"] = 1
        host = [tuple(x.numpy() if torch.is_tensor(x) else x for x in make_batch(4, 30 + (i % 2) * 4, seed=90 + i % 2, ragged=True)) for i in range(9)]
        bucket = (8, 32, int(c.model_config["max_seq_len"]))
        res = {}
        for mode in ("eager", "graph", "refused_once"):
            m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV, seed=5).train()      # dropout on: counters tick per step
            opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
            red = GradReducer(m.flat_buffers()[1], m.grad_buckets(8), m.group_offsets(), force_collectives=True)
            eng = TrainEngine(m, opt, c, FastSpeech2Loss(c.preprocess_config, c.model_config), reducer=red, hip_graph=mode != "eager")
            keep = G.GraphedTrainStep
            refused = [0]
            if mode == "refused_once":
                class RefuseFirst(keep):
                    def __init__(self, enqueue, example_batch, warmup=2, pool=None, **kw):
                        if refused[0] == 0:
                            refused[0] += 1
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                                enqueue(list(example_batch))             # a whole step recorded, collectives included, nothing executed
                            raise RuntimeError("capture refused (test)")
                        keep.__init__(self, enqueue, example_batch, warmup=warmup, pool=pool, **kw)
                E.GraphedTrainStep = RefuseFirst
            try:
                step = 0
                for b in DeviceFeeder(host, DEV, bucket=bucket):
                    step += 1
                    eng.step(b, step)
                torch.cuda.synchronize()
            finally:
                E.GraphedTrainStep = keep
            res[mode] = (list(red.history), m.flat_buffers()[0].cpu().clone(), dict(eng.stats), opt.current_step, opt._host_step)
        every = tuple(red.buckets)
        for mode, (hist, _, stats, cur, host_step) in res.items():
            assert len(hist) == 9 and all(h == every for h in hist), (mode, [len(h) for h in hist])
            assert cur == 9 and host_step == 9, (mode, cur, host_step)
        assert res["graph"][2]["replayed"] >= 4 and res["refused_once"][2].get("capture_failed", 0) == 1 and res["refused_once"][2]["replayed"] >= 2
        assert torch.equal(res["eager"][1], res["graph"][1]) and torch.equal(res["eager"][1], res["refused_once"][1])
    finally:
        if created:
            dist.destroy_process_group()
