"""Data-parallel training over the GPUs of one node: one process per GPU, gradients averaged with RCCL all-reduce
over xGMI on a side HIP stream while backward is still running.

The reference has no distributed code (only a commented-out nn.DataParallel, train.py:104); semantics here are
"N independent reference micro-batches, gradients averaged" = the reference's own gradient accumulation with
grad_acc_step = N (train.py:43-47), which is what the gloo test checks.

Design for xGMI (point-to-point links, per-link bound): the gradients live in ONE flat fp32 buffer laid out in
forward order, so buckets are contiguous slices that complete from the END of the buffer as backward proceeds;
each bucket is one large all-reduce (default 24 MB) issued as soon as its last gradient is written.  The 1/N
factor is folded into the loss gradient (grad_scale), so the collective is a plain SUM.
"""
import torch
import torch.distributed as dist


def dist_timeout_s(timeout_s=None):
    """The bound on every collective of the job, in seconds: the argument, else TTSK_DIST_TIMEOUT_S, else 300."""
    import os
    if timeout_s is None:
        timeout_s = float(os.environ.get("TTSK_DIST_TIMEOUT_S", "300"))
    return max(1.0, float(timeout_s))


def init_distributed(backend=None, timeout_s=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run contract).  backend 'nccl' is RCCL.
    Every collective of the group is BOUNDED (`timeout_s`, default 300 s, TTSK_DIST_TIMEOUT_S): a rank that waits longer for its peers
    — one of them died, or issued a different sequence of collectives — fails instead of hanging, and the job ends with a non-zero
    status (launch.spawn_ranks / torch.distributed.run then end the other ranks).  With RCCL the watchdog thread enforces it
    (TORCH_NCCL_ASYNC_ERROR_HANDLING=1: abort the communicator and the process); with gloo the collective raises.  A rank is never
    re-executed."""
    import datetime
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=dist_timeout_s(timeout_s)))
    return rank, world, local


class GradReducer:
    """Bucketed all-reduce of a flat gradient buffer.

    `flat_grad`: the fp32 buffer; `buckets`: [(start, end)] from the END of the buffer (FastSpeech2.grad_buckets);
    `group_offsets`: name -> lowest flat offset of the parameter group whose completion `on_group_done(name)`
    announces (groups complete in reverse forward order, so everything at or above that offset is final)."""

    def __init__(self, flat_grad, buckets, group_offsets, process_group=None, force_collectives=False, host_staged=None):
        self.flat_grad = flat_grad
        self.buckets = list(buckets)
        self.group_offsets = dict(group_offsets)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = bool(force_collectives) and dist.is_initialized()    # issue the all-reduces even at world size 1 (bench: DP schedule cost)
        # A process group that cannot take device tensors directly (gloo: two ranks sharing ONE GPU, which RCCL refuses — the only
        # cross-process run a 1-GPU box allows, tests/test_00_two_ranks_one_gpu.py): each bucket goes through a pinned host buffer —
        # an event behind its last writer, D2H, the collective on the host copy, H2D on the step's stream.  Same interface, same
        # bucket order on every rank, no overlap with backward; not capturable (the host waits inside `finish`).
        if host_staged is None:
            host_staged = dist.is_initialized() and flat_grad.is_cuda and dist.get_backend(process_group) == "gloo"
        self.host_staged = bool(host_staged)
        self._host = None
        self._staged = []           # (start, end, event recorded behind the bucket's last writer)
        self._next = 0
        self._handles = []
        self.launched = []          # (start, end) in launch order, for tests / tracing
        # One entry per finished step: the buckets whose all-reduce that step issued, in order.  Ranks must agree on this sequence step
        # by step — a rank that issues a different one leaves its peers waiting in a collective (bounded: init_distributed) — whatever
        # mix of replayed graphs and plain launches each of them runs (TrainEngine keeps it for replays too).
        self.history = []
        self._step_start = 0

    def _launch_down_to(self, watermark):
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= watermark:
            s, e = self.buckets[self._next]
            if self.world > 1 or self.force:
                if self.host_staged:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("a host-staged gradient reducer cannot be captured in a hipGraph")
                    ev = torch.cuda.Event()
                    ev.record()          # on the stream that announced the bucket: everything that wrote it is ahead of this
                    self._staged.append((s, e, ev))
                else:
                    self._handles.append(dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            self.launched.append((s, e))
            self._next += 1

    def _finish_staged(self):
        if self._host is None:
            self._host = torch.empty(self.flat_grad.numel(), dtype=self.flat_grad.dtype).pin_memory()
        for s, e, ev in self._staged:
            ev.synchronize()
            self._host[s:e].copy_(self.flat_grad[s:e])                       # D2H (synchronous for a pinned destination + sync below)
            torch.cuda.current_stream().synchronize()
            dist.all_reduce(self._host[s:e], op=dist.ReduceOp.SUM, group=self.pg)
            self.flat_grad[s:e].copy_(self._host[s:e], non_blocking=True)    # H2D on the step's stream: clip + Adam queue behind it
        torch.cuda.current_stream().synchronize()                            # the pinned buffer is reused by the next step
        self._staged = []

    def ready(self, name):
        """Would `on_group_done(name)` launch a bucket?  backward_native flushes its deferred weight-gradient work (one grouped
        launch + reducers) only then: with 24 MB buckets 6 of the 13 group boundaries, instead of a flush at every one."""
        return self._next < len(self.buckets) and self.buckets[self._next][0] >= self.group_offsets[name]

    def on_group_done(self, name):
        self._launch_down_to(self.group_offsets[name])

    def finish(self):
        """Issue whatever is left and make the current stream wait for every collective (no host block on NCCL)."""
        self._launch_down_to(0)
        if self._staged:
            self._finish_staged()
        for h in self._handles:
            h.wait()
        self._handles = []
        self._next = 0
        self.history.append(tuple(self.launched[self._step_start:]) if (self.world > 1 or self.force) else ())
        self._step_start = len(self.launched)

    def reset(self):
        """Forget a half-issued step (a hipGraph capture of it was aborted): no handles, first bucket next."""
        self._handles = []
        self._staged = []
        self._next = 0
        del self.launched[self._step_start:]

    def bucket_bytes(self):
        """Bytes of each bucket's all-reduce, in launch order."""
        return [int((e - s) * self.flat_grad.element_size()) for s, e in self.buckets]

    def grad_scale(self, grad_acc_step=1):
        return 1.0 / (grad_acc_step * self.world)
