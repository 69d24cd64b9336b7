"""CPU, world_size 2, gloo: the data-parallel gradient path (tts_king_amd/parallel.py) — bucketed all-reduce of the
model's flat gradient buffer, launched bucket by bucket as backward announces finished parameter groups.

The kernels need a GPU, so the gradients here are synthetic (rank-dependent fills); what is checked is the N>1 logic:
every rank launches the same buckets in the same order, every gradient element is reduced exactly once, the result is
the SUM over ranks (the 1/N is folded into the loss gradient through `grad_scale`), and a second step reuses the
reducer.  The equivalence "N ranks x one micro-batch == the reference's grad_acc_step = N" (train.py:43-47) is checked
with the oracle on two micro-batches."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    torch.set_num_threads(1)
    from tts_king_amd.config import default_config
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.parallel import GradReducer, init_distributed
    r, w, _ = init_distributed("gloo")
    assert (r, w) == (rank, world)
    cfg = default_config()
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device="cpu")
    flat, grad, _ = m.flat_buffers()
    n = grad.numel()
    red = GradReducer(grad, m.grad_buckets(8), m.group_offsets())
    from tts_king_amd.fastspeech2 import _GroupNotifier
    order = m.backward_group_order()            # the order backward_native itself enforces (its notifier raises on any other)
    offs = m.group_offsets()
    results = []
    flush_log = []
    for step in range(2):
        base = torch.arange(n, dtype=torch.float32) % 1000
        grad.zero_()
        # backward_native's real notifier: gradients of a group become visible in the flat buffer only at a FLUSH (they sit in
        # the deferred grouped-GEMM queue until then), and the notifier flushes only when the reducer says a bucket completes
        pending = []

        def flush():
            for lo, hi in pending:
                grad[lo:hi] = base[lo:hi] * (rank + 1) + step
            pending.clear()
        notifier = _GroupNotifier(order, red.on_group_done, flush)
        watermarks = []
        hi = n
        for name in order:
            pending.append((offs[name], hi))     # this group's gradients are queued now
            hi = offs[name]
            notifier.done(name)
            watermarks.append(red.buckets[red._next - 1][0] if red._next else n)
        flush()
        flush_log.append(notifier.flushes)
        launched_before_finish = list(red.launched)
        red.finish()
        want = base * sum(range(1, world + 1)) + step * world
        results.append((bool(torch.equal(grad, want)), launched_before_finish, list(red.launched)))
        red.launched.clear()
    # a bucket is only launched once every gradient in it is final: its start is >= the announced group's offset
    ok_order = all(s >= offs[name] for name, s in zip(order, watermarks))
    ok_order = ok_order and all(f < len(order) for f in flush_log) and flush_log[0] == flush_log[1] >= 2     # fewer flushes than groups
    q.put((rank, results, ok_order, red.grad_scale(1), n))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    out.sort()
    (_, res0, ok0, gs0, n), (_, res1, ok1, gs1, _) = out
    assert ok0 and ok1 and gs0 == gs1 == 0.5
    for (eq0, pre0, all0), (eq1, pre1, all1) in zip(res0, res1):
        assert eq0 and eq1                                   # SUM over ranks, every element
        assert all0 == all1 and pre0 == pre1                 # same buckets, same order on both ranks
        assert len(pre0) >= len(all0) - 1                    # overlap: (almost) everything is in flight before finish()
        covered = sorted(all0)
        assert covered[0][0] == 0 and covered[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))      # exactly once
        assert all0 == sorted(all0, key=lambda se: -se[0])   # launched from the END of the buffer (backward order)


def test_dp_equals_reference_grad_accumulation(cfg):
    """N ranks x one micro-batch with gradients scaled by 1/N and summed == the reference's gradient accumulation over
    the same N micro-batches (train.py:43-47: `(total_loss / grad_acc_step).backward()` N times, then one step)."""
    import copy
    from oracle import fs2 as ofs2
    from tests.oracle_util import fs2_state_dict
    from tts_king_amd.synthetic import make_batch
    torch.manual_seed(0)
    c = copy.deepcopy(cfg)
    mc = c.model_config
    mc["transformer"]["encoder_layer"], mc["transformer"]["decoder_layer"] = 1, 1      # small: the property is structural
    sd = {k: v for k, v in fs2_state_dict(c, 5).items() if "layer_stack" not in k or ".0." in k}
    keep = ofs2._drop
    ofs2._drop = lambda x, p, train: x
    try:
        N = 2
        batches = [make_batch(2, 12, seed=40 + r, ragged=True) for r in range(N)]
        # reference semantics: accumulate N micro-batches with loss / N
        c.train_config["optimizer"]["grad_acc_step"] = N
        tr = ofs2.OracleTrainer(sd, mc, c.train_config, 0)
        for i, b in enumerate(batches):
            out = ofs2.fs2_forward(tr.sd, mc, *b[2:], train=True, bn_buffers={})
            (ofs2.fs2_loss(b, out)[0] / N).sum().backward()
        acc = {k: tr.sd[k].grad.clone() for k in tr.keys if tr.sd[k].grad is not None}
        # DP semantics: each rank computes grad_scale = 1/N times its own micro-batch gradient; all-reduce SUM
        summed = None
        for b in batches:
            t2 = ofs2.OracleTrainer(sd, mc, c.train_config, 0)
            out = ofs2.fs2_forward(t2.sd, mc, *b[2:], train=True, bn_buffers={})
            (ofs2.fs2_loss(b, out)[0] * (1.0 / N)).sum().backward()
            g = {k: t2.sd[k].grad for k in t2.keys if t2.sd[k].grad is not None}
            summed = g if summed is None else {k: summed[k] + g[k] for k in g}
        for k in acc:
            assert torch.allclose(acc[k], summed[k], rtol=1e-5, atol=1e-7), k
    finally:
        ofs2._drop = keep
