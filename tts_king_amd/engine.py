"""The train step as the trainer runs it: shape-bucketed batches, one replayed hipGraph per shape, data-parallel reducer.

reference: train.py:78-235 (`main`: the loop around `main_train_step`).  The reference pads every batch to its own longest
utterance, so almost every step has a new (L, T); a hipGraph needs static shapes.  Here a batch is padded a little further,
to the next multiple of `l_bucket` phonemes and `t_bucket` frames (`pad_to_bucket`, host side, before the pinned H2D copy),
and the sizes the reference would have seen travel along as device scalars.  Most of the model does not notice the extra PAD
positions (PAD rows are zeroed after every sub-layer, keys are masked, the phoneme losses select valid positions); the places
that would — PostNet BatchNorm statistics, the PostNet convs' zero padding, the mel-loss denominators (`frame_limit`), and the
VariancePredictor convs, whose inputs carry the speaker / pitch embeddings at PAD positions too (`phoneme_limit`) — treat
positions past the batch's own longest utterance as non-existent (include/ttsk.h: BatchNorm, loss, va_embed).  So a bucketed step computes
what the reference computes on the unpadded batch, and shapes repeat: one graph per (B, L_bucket, T_bucket, update?).

A shape runs eagerly the first time it is seen and is captured the second time; graphs share one memory pool and are kept
in an LRU list (`mi355x.max_graphs`).  Tensors returned by a replay (the model outputs) live in that pool: they are valid
until the next step.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .graph import GraphedTrainStep, make_enqueue


class PaddedBatch(tuple):
    """The 15-tuple batch plus `t_true` / `l_true` (frames / phonemes of the batch's own longest utterance; None when the bucket adds
    no padding) and, on the device, `frame_limit` (int32[1]) / `phoneme_limit` (int64[B], every entry l_true).  From the DeviceFeeder
    every device field (the limits included) is a view of ONE buffer, `packed`, laid out as `layout` says (dataset.views_of)."""
    t_true = None
    l_true = None
    frame_limit = None
    phoneme_limit = None
    packed = None
    layout = None


def bucket_up(n, step):
    return (int(n) + step - 1) // step * step


def bucket_plan(batch, l_bucket=8, t_bucket=32, max_seq_len=1000):
    """What `pad_to_bucket` will do to a numpy 15-tuple, without doing it: (Lb, Tb, t_true, l_true, {field index: padded length of its
    axis 1}).  Batches longer than `max_seq_len` frames keep their length (the train-mode decoder truncation path, Models.py:172-180)."""
    max_src, max_mel = batch[5], batch[8]
    Lb = bucket_up(max_src, l_bucket)
    if Lb > max_seq_len >= int(max_src):
        Lb = max(int(max_src), max_seq_len)        # never pad a text past the position table (the eval-only long-input path)
    t_true = int(max_mel)
    Tb = bucket_up(t_true, t_bucket) if t_true <= max_seq_len else t_true
    if Tb > max_seq_len >= t_true:
        Tb = max(t_true, max_seq_len)
    frame_level_p = np.asarray(batch[11]).shape[1] == int(max_mel) and int(max_mel) != int(max_src)
    frame_level_e = np.asarray(batch[9]).shape[1] == int(max_mel) and int(max_mel) != int(max_src)
    axis1 = {3: Lb, 6: Tb, 9: Tb if frame_level_e else Lb, 10: Lb, 11: Tb if frame_level_p else Lb, 12: Lb}
    return Lb, Tb, (t_true if Tb != t_true else None), (int(max_src) if Lb != int(max_src) else None), axis1


def pad_to_bucket(batch, l_bucket=8, t_bucket=32, max_seq_len=1000):
    """numpy 15-tuple (fs_two/dataset.py:188-204 layout) -> PaddedBatch with texts / durations / pitch / energy padded to a
    multiple of `l_bucket` phonemes and mels to a multiple of `t_bucket` frames (zeros = the reference's PAD values)."""
    Lb, Tb, t_true, l_true, axis1 = bucket_plan(batch, l_bucket, t_bucket, max_seq_len)

    def pad(a, n):
        a = np.asarray(a)
        if a.shape[1] == n:
            return a
        w = [(0, 0)] * a.ndim
        w[1] = (0, n - a.shape[1])
        return np.pad(a, w, mode="constant", constant_values=0)

    fields = list(batch)
    for i, n in axis1.items():
        fields[i] = pad(fields[i], n)
    fields[5], fields[8] = Lb, Tb
    out = PaddedBatch(fields)
    out.t_true, out.l_true = t_true, l_true
    return out


class TrainEngine:
    """`step(batch, step_no)` = the device work of `main_train_step` (train.py:24-56), replayed from a hipGraph when the batch's
    shape has been seen before.  Returns (losses fp32[8] on the device: total, mel, pitch, energy, duration, 0, 0, n_valid —
    NOT divided by grad_acc_step, the caller does that when it reads them; the model's 9-tuple of outputs)."""

    def __init__(self, model, optimizer, cfg, Loss, reducer=None, hip_graph=None):
        mi = cfg.get("mi355x", {}) if hasattr(cfg, "get") else {}
        self.model, self.optimizer, self.cfg, self.Loss, self.reducer = model, optimizer, cfg, Loss, reducer
        self.use_graph = bool(mi.get("hip_graph", True)) if hip_graph is None else bool(hip_graph)
        self.max_graphs = int(mi.get("max_graphs", 64))
        self.grad_acc = int(cfg.train_config["optimizer"]["grad_acc_step"])
        self.grad_scale = reducer.grad_scale(self.grad_acc) if reducer is not None else None
        self._graphs = OrderedDict()
        self._seen = set()
        self._eager_only = set()          # shapes whose capture failed: plain launches from then on
        self._pool = None
        # collectives inside a replayed graph are invisible to the process group's timeout and to RCCL's watchdog: the host bounds them
        self.guard = None
        if reducer is not None and (getattr(reducer, "world", 1) > 1 or getattr(reducer, "force", False)):
            from .parallel import ReplayGuard
            self.guard = ReplayGuard()
        self.stats = {"eager": 0, "captured": 0, "replayed": 0}
        from .hostcpu import fit_torch_threads
        fit_torch_threads()       # host-side staging copies on the cores this process is granted, not on every core the host shows

    @staticmethod
    def _key(batch, is_update, limited):
        return tuple(tuple(t.shape) for t in batch if torch.is_tensor(t)) + (int(batch[5]), int(batch[8]), bool(is_update), limited)

    def _enqueue(self, is_update, frame_limit, phoneme_limit, accumulate=None):
        return make_enqueue(self.model, self.optimizer, self.cfg, self.Loss, step_is_update=is_update,
                            reducer=self.reducer if is_update else None, grad_scale=self.grad_scale, frame_limit=frame_limit,
                            phoneme_limit=phoneme_limit, accumulate=accumulate)

    def step(self, batch, step_no):
        is_update = step_no % self.grad_acc == 0
        fl, pl = getattr(batch, "frame_limit", None), getattr(batch, "phoneme_limit", None)
        if not self.model.training:
            self.model.train()             # (nn.Module.train() walks every sub-module: 7 ms here, so not once per step)
        # longer than the position tables (frames: the train-mode truncation path; phonemes: sinusoid_table(...).to(dev) is a pageable
        # H2D copy, not capturable, and the grouped predictor path with its phoneme_limit is off there): plain launches
        # ... and so does an update step whose gradient reducer goes through the host (parallel.GradReducer.host_staged: the host
        # waits for the collective inside the step); the accumulate-only micro-steps of such a run still replay
        staged = is_update and self.reducer is not None and getattr(self.reducer, "host_staged", False)
        if not self.use_graph or staged or int(batch[8]) > self.model.max_seq_len or int(batch[5]) > self.model.max_seq_len:
            self.stats["eager"] += 1
            losses, out = self._enqueue(is_update, fl, pl)(batch)
            return losses, out
        # first micro-step after an update overwrites the gradient buffer, later ones accumulate: part of a captured graph's identity
        acc = bool(self.model.grads_partial)
        key = self._key(batch, is_update, (fl is not None, pl is not None, acc))
        g = self._graphs.get(key)
        if key in self._eager_only:
            self.stats["eager"] += 1
            return self._enqueue(is_update, fl, pl)(batch)
        if g is None and key not in self._seen:
            self._seen.add(key)                       # first sight of a shape: plain launches (lazy allocations, split-K plans)
            self.stats["eager"] += 1
            return self._enqueue(is_update, fl, pl)(batch)
        if g is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.popitem(last=False)
            static_packed = None
            if getattr(batch, "packed", None) is not None:
                # the graph's static inputs as views of one buffer with the feeder's layout: a replay takes its batch with ONE copy
                from .dataset import views_of
                static_packed = batch.packed.clone()
                views = views_of(static_packed, batch.layout)
                static = [views[i] if i in views else t for i, t in enumerate(batch)]
                static_fl, static_pl = views.get("fl"), views.get("pl")
            else:
                static = [t.clone() if torch.is_tensor(t) else t for t in batch]
                static_fl = fl.clone() if fl is not None else None
                static_pl = pl.clone() if pl is not None else None
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            host_step = self.optimizer._host_step
            if self.reducer is not None:
                self.reducer.trim()        # (no trim inside the capture's finish(): n_hist stays a valid index)
            n_hist = len(self.reducer.history) if self.reducer is not None else 0
            torch.cuda.synchronize()
            try:
                g = GraphedTrainStep(self._enqueue(is_update, static_fl, static_pl, accumulate=acc), static, warmup=0, pool=self._pool,
                                     static_packed=static_packed, layout=getattr(batch, "layout", None))
            except Exception as e:      # e.g. RCCL refusing to have a collective captured: this shape runs with plain launches
                # the aborted capture ran part of one step's Python: put the host-side bookkeeping back where it was
                self.optimizer._host_step = host_step
                if self.reducer is not None:
                    self.reducer.reset()
                    del self.reducer.history[n_hist:]       # (the aborted capture's Python may have closed a step that never ran)
                self.model.abort_step()
                self.model.grads_partial = acc
                torch.cuda.synchronize()
                self._eager_only.add(key)
                self.stats["capture_failed"] = self.stats.get("capture_failed", 0) + 1
                self.last_capture_error = "%s: %s" % (type(e).__name__, e)
                self.stats["eager"] += 1
                return self._enqueue(is_update, fl, pl)(batch)
            self.optimizer._host_step = host_step     # capture ran the Python of one step without executing it
            # ... including the reducer's: the collectives it recorded are what every replay of this graph will execute
            g.collectives = ()
            if self.reducer is not None:
                g.collectives = self.reducer.history[n_hist] if len(self.reducer.history) > n_hist else ()
                del self.reducer.history[n_hist:]
            g.frame_limit, g.phoneme_limit = static_fl, static_pl
            self._graphs[key] = g
            self.stats["captured"] += 1
        else:
            self._graphs.move_to_end(key)
            self.stats["replayed"] += 1
        if not g.takes_packed(batch):       # (a packed batch carries its limits inside the one buffer g.run copies)
            if g.frame_limit is not None:
                g.frame_limit.copy_(fl, non_blocking=True)
            if g.phoneme_limit is not None:
                g.phoneme_limit.copy_(pl, non_blocking=True)
        guarded = self.guard is not None and is_update and len(g.collectives) > 0
        if guarded:
            self.guard.before_step()       # the replay two back must be done: bounds a captured all-reduce whose peer is gone
        losses, out = g.run(batch)
        if guarded:
            self.guard.after_step()
        if self.reducer is not None and is_update:
            self.reducer.history.append(g.collectives)
            if len(self.reducer.history) > 2 * self.reducer.KEEP:
                self.reducer.trim()
        self.model.grads_partial = not is_update      # what the replayed step's Python would have left
        if is_update:
            self.optimizer._host_step += 1
        return losses.clone(), out

    def wait(self):
        """Everything enqueued so far has finished — within the data-path bound when replayed steps with collectives are outstanding
        (parallel.ReplayGuard; raises CollectiveTimeout), then a plain synchronize.  The trainer calls this before every host read."""
        if self.guard is not None:
            self.guard.wait_all()
        torch.cuda.synchronize()
