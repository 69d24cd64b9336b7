"""FS2 trainer with the reference's entry points (reference: train.py:24-56 `main_train_step`, :78-235 `main`,
fs_two/evaluate.py:18-101 `evaluate`), running the MI355X train step.

What runs each step is `tts_king_amd.engine.TrainEngine` (shape buckets + hipGraph replay + data-parallel reducer) — the
same device work `main_train_step` enqueues.  Kept from the reference: the step function and its return value, the loop structure (DataLoader over groups of
batch_size*4 utterances sorted by phoneme count and cut into batches, `grad_acc_step`, log / val / save cadence from
`train_config.step`), the validation message, the checkpoint layout `{"model", "embedding", "optimizer"}`
(train.py:212-227).  Different: batches reach the GPU through a pinned-memory prefetcher (tts_king_amd/dataset.py),
wandb / matplotlib logging is replaced by a plain print (no network on the GPU box), and resuming also
restores the optimizer state the reference saves but never reloads.
"""
import os
import sys

import torch
from torch.utils.data import DataLoader

from tts_king_amd.config import load_config
from tts_king_amd.dataset import Dataset, DeviceFeeder
from tts_king_amd.loss import FastSpeech2Loss
from tts_king_amd.train_step import get_model, main_train_step, save_checkpoint, to_device  # noqa: F401  (reference names)


def get_param_num(model):
    """reference: fs_two/utils/model.py:41-43."""
    return sum(p.numel() for p in model.parameters())


def validation_shard(n_items, batch_size, rank=0, world=1):
    """The reference's validation batches (fs_two/evaluate.py:30-36: consecutive groups of `batch_size` items of the unshuffled set,
    the last one short) dealt round-robin to the ranks: rank r gets batches r, r + world, ...  Batch COMPOSITION is the reference's —
    the mel losses are means over a batch's padded frames, so regrouping the utterances would change the numbers — only who
    evaluates a batch changes.  Returns the list of index lists this rank evaluates."""
    batches = [list(range(i, min(i + batch_size, n_items))) for i in range(0, n_items, batch_size)]
    return batches[rank::world]


def evaluate(model, step, cfg, logger=None, train_val="val", vocoder=None, device=0, ctrl=None):
    """reference: fs_two/evaluate.py:18-101 — teacher-forced forward in eval mode over `val.txt`, loss means weighted by
    batch size over len(dataset); returns the reference's message string (logger output is not reproduced).
    `ctrl` (parallel.ControlPlane of a multi-rank job): every rank evaluates its share of the reference's batches
    (`validation_shard`) and the weighted sums are added over the control-plane group — no rank waits for rank 0 to walk the whole
    set alone, and the wait for the slowest shard runs against the control plane's bound, not the gradient all-reduces' (VERDICT r05
    item 11).  Every rank returns the same message."""
    dataset = Dataset("%s.txt" % train_val, cfg.preprocess_config, cfg.train_config, sort=False, drop_last=False)
    batch_size = cfg.train_config["optimizer"]["batch_size"]
    rank, world = (ctrl.rank, ctrl.world) if ctrl is not None else (0, 1)
    loader = DataLoader(dataset, batch_sampler=validation_shard(len(dataset), batch_size, rank, world), collate_fn=dataset.collate_fn)
    Loss = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)
    dev = model.device
    was_training = model.training
    model.eval()
    loss_sums = [0.0 for _ in range(6)]
    with torch.no_grad():
        for batchs in loader:
            for batch in DeviceFeeder(batchs, dev):
                output = model(*(batch[2:]))
                losses = Loss(batch, output)
                vals = torch.stack([l.reshape(-1)[0].float() for l in losses[1:]]).cpu().tolist()   # one host read per batch
                for i, v in enumerate(vals):
                    loss_sums[i] += v * len(batch[0])
    model.train(was_training)
    if ctrl is not None:
        loss_sums = ctrl.all_reduce_sums(loss_sums)
    loss_means = [s / len(dataset) for s in loss_sums]
    loss_means = [sum(loss_means)] + loss_means
    return """Validation Step {}, 
                 Total Loss: {:.4f}, 
                 Mel Loss: {:.4f}, 
                 Pitch Loss: {:.4f}, 
                 Mean pitch {:.4f},
                 Std pitch {:.4f}""".format(step, *loss_means[:5])


def load_training_state(model, optimizer, path):
    """Resume from a checkpoint written by save_checkpoint: weights, speaker embedding (re-inserted as in fsapi.py:28-30)
    and — unlike the reference, which saves it but has no caller for ScheduledOptim.load_state_dict — the Adam state."""
    ckpt = torch.load(path, map_location="cpu")
    state = dict(ckpt["model"])
    state["speaker_emb.weight"] = ckpt["embedding"]
    model.load_state_dict(state)
    if optimizer is not None and isinstance(ckpt.get("optimizer"), dict):
        optimizer.load_state_dict(ckpt["optimizer"])       # flat layout or the reference's torch.optim.Adam layout


def main(cfg, max_steps=None):
    """reference: train.py:78-235.  `max_steps` (extra) stops early instead of the reference's `quit()` at total_step.

    What runs per step is tts_king_amd.engine.TrainEngine: batches padded to shape buckets on the host, one replayed hipGraph
    per shape (`mi355x.hip_graph`), and — under `python -m torch.distributed.run --nproc-per-node N train.py` — one process per
    GPU, each on its own shard of every epoch's shuffle, gradients all-reduced bucket by bucket over RCCL while backward runs
    (tts_king_amd.parallel).  Losses are read from the device only when they are logged."""
    from tts_king_amd.engine import TrainEngine
    from tts_king_amd.parallel import ControlPlane, GradReducer, init_distributed
    print("Prepare training ...")
    rank, world, local = init_distributed()
    ctrl = ControlPlane()          # validation sums and the wait for rank 0's checkpoint write: a group and a bound of their own
    device = "cuda:%d" % local if world > 1 else cfg.gpu
    dataset = Dataset("train.txt", cfg.preprocess_config, cfg.train_config, sort=True, drop_last=True)
    batch_size = cfg.train_config["optimizer"]["batch_size"]
    group_size = 4                                    # sorting happens inside groups of 4 batches (train.py:91)
    assert batch_size * group_size < len(dataset)
    mi = cfg.get("mi355x", {}) if hasattr(cfg, "get") else {}
    workers = int(mi.get("loader_workers", 4)) if mi else 4          # reference: num_workers=4 (train.py:98)
    sampler = None
    if world > 1:       # every rank draws its own 1/N of each epoch's permutation (same seed on all ranks, disjoint indices)
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=int(mi.get("seed", 1234)), drop_last=True)
    loader = DataLoader(dataset, batch_size=batch_size * group_size, shuffle=sampler is None, sampler=sampler,
                        collate_fn=dataset.collate_fn, num_workers=workers)
    model, optimizer = get_model(cfg, device, train=True)
    Loss = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)
    reducer = None
    if world > 1:
        # Every rank builds the same weights from the same seed, but "N independent reference micro-batches" draw N different
        # dropout masks: the Philox key of rank r is (rank 0's key) + r.  A checkpoint stores rank 0's key, so the same rule
        # applies after a resume.
        optimizer.state[2] += rank
        reducer = GradReducer(model.flat_buffers()[1], model.grad_buckets(mi.get("dp_bucket_mb", 24)), model.group_offsets())
    engine = TrainEngine(model, optimizer, cfg, Loss, reducer=reducer)
    bucket = None
    if bool(mi.get("bucket_shapes", True)):
        bucket = (int(mi.get("l_bucket", 8)), int(mi.get("t_bucket", 32)), int(cfg.model_config["max_seq_len"]))
    if rank == 0:
        print("Number of FastSpeech2 Parameters:", get_param_num(model))
        for p in cfg.train_config["path"].values():
            os.makedirs(p, exist_ok=True)
    step = cfg.tts.restore_step + 1
    total_step = cfg.train_config["step"]["total_step"]
    if max_steps is not None:
        total_step = min(total_step, cfg.tts.restore_step + max_steps)
    st = cfg.train_config["step"]
    grad_acc = cfg.train_config["optimizer"]["grad_acc_step"]
    epoch = 1
    model.engine = engine
    while True:
        if sampler is not None:
            sampler.set_epoch(epoch)
        for batchs in loader:
            for batch in DeviceFeeder(batchs, model.device, bucket=bucket):
                losses, output = engine.step(batch, step)
                if rank == 0 and step % st["log_step"] == 0:
                    engine.wait()                                                     # (bounded when replayed collectives are outstanding)
                    vals = [v / grad_acc for v in losses.cpu().tolist()[1:5]]         # the loop's only host read
                    print("Step {}/{}, Total Loss: {:.4f}, Mel Loss: {:.4f}, Pitch Loss: {:.4f}, Energy Loss: {:.4f}, "
                          "Duration Loss: {:.4f}".format(step, total_step, sum(vals), *vals))
                # Validation and checkpoint: no rank may run ahead into the next step's gradient all-reduce while another is still busy
                # here (that wait would run against the data-path bound and abort the job).  Every rank validates its share of the
                # batches; rank 0 alone writes the checkpoint while the others wait at the control plane's barrier.
                if step % st["val_step"] == 0:
                    engine.wait()
                    msg = evaluate(model, step, cfg, None, "val", None, device, ctrl=ctrl if world > 1 else None)
                    if rank == 0:
                        print(msg)
                if step % st["save_step"] == 0:
                    engine.wait()
                    if rank == 0:
                        save_checkpoint(model, optimizer, os.path.join(cfg.train_config["path"]["ckpt_path"], "{}.pth.tar".format(step)))
                    ctrl.barrier()
                if step == total_step:
                    engine.wait()
                    return model, optimizer
                step += 1
        epoch += 1


if __name__ == "__main__":
    _cfg = load_config("./config.yaml")
    # `mi355x.gpus: N` (N > 1) without a torch.distributed.run wrapper: this process starts its own N ranks — before anything here
    # touches the GPU — and only relays their output (tts_king_amd/launch.py); under torch.distributed.run the rank environment is
    # already there and `main` runs as one of the ranks.
    from tts_king_amd import launch as _launch
    _n = int((_cfg.get("mi355x", {}) or {}).get("gpus", 1)) if hasattr(_cfg, "get") else 1
    if _launch.wants_spawn(_n):
        raise SystemExit(_launch.spawn_ranks(_n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    from tts_king_amd.parallel import CollectiveTimeout as _CollectiveTimeout
    try:
        main(_cfg)
    except _CollectiveTimeout as _e:
        # the device queue is stuck behind a collective whose peer is gone: a normal interpreter exit would wait for it in the
        # teardown of the process group.  Say why and leave at once, non-zero; the launcher ends the other ranks.
        sys.stderr.write("train.py: %s\n" % _e)
        sys.stderr.flush()
        os._exit(75)
