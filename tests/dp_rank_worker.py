"""One rank of tests/test_00_two_ranks_one_gpu.py (started by tts_king_amd.launch.spawn_ranks; not a test module itself).

Every rank uses cuda:0 (a 1-GPU box): RCCL refuses two ranks on one device, so the process group is gloo and the gradient reducer
stages its buckets through pinned host memory (parallel.GradReducer.host_staged) — the same GradReducer / _GroupNotifier /
TrainEngine code the RCCL path runs, with real cross-process collectives.  usage: dp_rank_worker.py <outdir> <n_updates>"""
import copy
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_batches(rank, n_micro):
    """Per-rank micro-batches (micro-step i = 2k is accumulate-only, i = 2k + 1 the update).  The accumulate-only ones recur so that
    graphs are captured and replayed, at DIFFERENT steps on the two ranks: rank 0 alternates A, B, A, B, ... (A: eager at k = 0,
    captured at k = 2, replayed at k = 4; B one step later), rank 1 runs C, C, C, D, D, D (captured at k = 1 and k = 4, replayed at
    k = 2 and k = 5) — so one rank replays a graph while the other captures or launches eagerly.  Different batch sizes and lengths
    per rank; the update micro-steps (eager: the reducer goes through the host) use further shapes."""
    from tts_king_amd.synthetic import make_batch
    out = []
    for i in range(n_micro):
        k, upd = i // 2, i % 2 == 1
        if rank == 0:
            L, seed = ((32, 500) if k % 2 == 0 else (40, 501)) if not upd else (48, 502)
        else:
            L, seed = ((36, 600) if (k // 3) % 2 == 0 else (44, 601)) if not upd else ((28, 602) if k % 2 == 0 else (52, 603))
        out.append(make_batch(3 + rank, L, seed=seed, ragged=True))
    return out


def build(cfg, dev):
    from tests.oracle_util import fs2_state_dict
    from tts_king_amd.fastspeech2 import FastSpeech2
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=dev)
    m.load_state_dict(fs2_state_dict(cfg, 7))
    m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0
    return m.train()


def main():
    outdir, n_updates = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from tts_king_amd.config import default_config
    from tts_king_amd.dataset import DeviceFeeder
    from tts_king_amd.engine import TrainEngine
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.parallel import GradReducer
    dev = "cuda:0"
    cfg = copy.deepcopy(default_config())
    cfg.train_config["optimizer"]["grad_acc_step"] = 2
    n_micro = 2 * n_updates
    m = build(cfg, dev)
    opt = ScheduledOptim(m, cfg.train_config, cfg.model_config, 0)
    red = GradReducer(m.flat_buffers()[1], m.grad_buckets(24), m.group_offsets())
    assert red.host_staged and red.world == world
    eng = TrainEngine(m, opt, cfg, FastSpeech2Loss(cfg.preprocess_config, cfg.model_config), reducer=red, hip_graph=True)
    host = [tuple(x.numpy() if torch.is_tensor(x) else x for x in b) for b in rank_batches(rank, n_micro)]
    bucket = (8, 32, int(cfg.model_config["max_seq_len"]))
    step = 0
    for b in DeviceFeeder(host, dev, bucket=bucket):
        step += 1
        eng.step(b, step)
    torch.cuda.synchronize()
    w = m.flat_buffers()[0].cpu().clone()
    torch.save({"weights": w, "stats": dict(eng.stats), "updates": opt.current_step}, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    verdict = None
    if rank == 0:
        # the same run as ONE process: every update = the two ranks' micro-batch pairs accumulated with the same 1 / (2 * 2) scale
        # (the reference's grad_acc_step = 4 over those four batches, train.py:43-54), through the same feeder (shape buckets with
        # their frame / phoneme limits) and the same step closure — minus the reducer.  A SUM of the two ranks' fp32 gradients is
        # commutative, so the weights must be identical bit for bit.
        from tts_king_amd.graph import make_enqueue
        other = torch.load(os.path.join(outdir, "rank1.pt"))
        m2 = build(cfg, dev)
        opt2 = ScheduledOptim(m2, cfg.train_config, cfg.model_config, 0)
        loss2 = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)
        fed = []
        for r in range(world):
            hb = [tuple(x.numpy() if torch.is_tensor(x) else x for x in b) for b in rank_batches(r, n_micro)]
            fed.append(list(DeviceFeeder(hb, dev, bucket=bucket)))
        g = m2.flat_buffers()[1]
        for u in range(n_updates):
            per_rank = []
            for r in range(world):
                g.zero_()
                for k in range(2):
                    b = fed[r][2 * u + k]
                    make_enqueue(m2, opt2, cfg, loss2, step_is_update=False, grad_scale=1.0 / (2 * world), frame_limit=getattr(b, "frame_limit", None),
                                 phoneme_limit=getattr(b, "phoneme_limit", None), accumulate=k > 0)(b)
                per_rank.append(g.clone())
            g.copy_(per_rank[0] + per_rank[1])
            opt2.step_and_update_lr(advance_rng=False, keep_grads=True)
        torch.cuda.synchronize()
        w2 = m2.flat_buffers()[0].cpu()
        verdict = {"ranks_equal": bool(torch.equal(w, other["weights"])), "equals_one_process": bool(torch.equal(w, w2)),
                   "max_abs_vs_one_process": float((w - w2).abs().max()), "stats": [dict(eng.stats), other["stats"]],
                   "updates": [int(opt.current_step), int(other["updates"])]}
        with open(os.path.join(outdir, "verdict.json"), "w") as f:
            json.dump(verdict, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
