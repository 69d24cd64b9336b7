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
    """The bound on every DATA-PATH collective of the job (the gradient all-reduces), in seconds: the argument, else
    TTSK_DIST_TIMEOUT_S, else 300.  Eagerly issued collectives are bounded by the process group itself (init_distributed); collectives
    inside a replayed hipGraph enqueue no Work object the RCCL watchdog could see, so the host bounds those (ReplayGuard)."""
    import os
    if timeout_s is None:
        timeout_s = float(os.environ.get("TTSK_DIST_TIMEOUT_S", "300"))
    return max(1.0, float(timeout_s))


def ctrl_timeout_s(timeout_s=None):
    """The bound on CONTROL-PLANE waits (ranks waiting for rank 0's checkpoint write, for the slowest rank's validation shard): the
    argument, else TTSK_CTRL_TIMEOUT_S, else 6 h.  Separate from — and much larger than — the data-path bound: a validation pass
    or a checkpoint write on a slow disk may legitimately take longer than any gradient all-reduce ever should."""
    import os
    if timeout_s is None:
        timeout_s = float(os.environ.get("TTSK_CTRL_TIMEOUT_S", str(6 * 3600)))
    return max(1.0, float(timeout_s))


class CollectiveTimeout(RuntimeError):
    """A rank waited longer than the bound for device work that contains a collective (a peer died or issued a different sequence).
    The job must end non-zero (train.py / bench.py exit; launch.spawn_ranks / torch.distributed.run then end the peers); never re-exec."""


class ControlPlane:
    """The job's slow, host-side synchronisation, on a process group of its own (gloo, host tensors) whose timeout is `ctrl_timeout_s`:
    `barrier()` where ranks wait for work only one of them does (rank 0 writes the checkpoint; reference: train.py:212-227), and
    `all_reduce_sums()` for validation sums computed shard by shard (reference: fs_two/evaluate.py:62-75 runs them on one process).
    While ranks wait here none of them has issued the next step's gradient all-reduce, so the data-path bound (`dist_timeout_s`, which
    a long validation pass on rank 0 used to exhaust: VERDICT r05 item 11) is not running.  At world size 1 every call returns at once."""

    def __init__(self, timeout_s=None):
        import datetime
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.timeout_s = ctrl_timeout_s(timeout_s)
        self.pg = None
        if self.world > 1:
            self.pg = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=self.timeout_s))   # (every rank calls this)

    def barrier(self):
        if self.pg is not None:
            dist.barrier(group=self.pg)

    def all_reduce_sums(self, values):
        """Element-wise SUM over the ranks of a list of Python floats (float64 on the host, fixed rank order inside gloo)."""
        if self.pg is None:
            return [float(v) for v in values]
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t.tolist()


class ReplayGuard:
    """Host-side bound for steps whose collectives were captured into a hipGraph (a replay enqueues no Work object, so
    neither the process-group timeout nor the RCCL watchdog covers it — a peer that died leaves the replaying rank's queue stuck
    forever).  `after_step()` records an event behind the replay; before the host enqueues step k it requires step k - `depth` to have
    finished, polling with the data-path bound, and `wait_all()` does the same for everything outstanding (before a host read of the
    losses, at the end of the run).  In the steady state the event of two steps ago has long completed: one `query()` per step.
    Raises CollectiveTimeout on expiry."""

    def __init__(self, timeout_s=None, depth=2, make_event=None, clock=None, sleep=None):
        import collections
        import time
        self.timeout_s = dist_timeout_s(timeout_s)
        self.depth = int(depth)
        self._events = collections.deque()
        self._make_event = make_event or (lambda: torch.cuda.Event())
        self._clock = clock or time.monotonic
        self._sleep = sleep or time.sleep

    def _wait(self, ev, what):
        t0 = self._clock()
        pause = 50e-6
        while not ev.query():
            if self._clock() - t0 > self.timeout_s:
                raise CollectiveTimeout("%s did not finish within %.0f s (TTSK_DIST_TIMEOUT_S): a replayed step's all-reduce is waiting for "
                                        "a peer that is gone or out of step" % (what, self.timeout_s))
            self._sleep(pause)
            pause = min(pause * 2, 0.05)

    def before_step(self):
        while len(self._events) >= self.depth:
            self._wait(self._events.popleft(), "the step %d replays back" % self.depth)

    def after_step(self):
        ev = self._make_event()
        ev.record()
        self._events.append(ev)

    def wait_all(self):
        while self._events:
            self._wait(self._events.popleft(), "an outstanding replayed step")


def init_distributed(backend=None, timeout_s=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run contract).  backend 'nccl' is RCCL.
    Every EAGERLY ISSUED collective of the group is bounded (`timeout_s`, default 300 s, TTSK_DIST_TIMEOUT_S): a rank that waits longer
    for its peers — one of them died, or issued a different sequence of collectives — fails instead of hanging, and the job ends with a
    non-zero status (launch.spawn_ranks / torch.distributed.run then end the other ranks).  With RCCL the watchdog thread enforces it
    (TORCH_NCCL_ASYNC_ERROR_HANDLING=1: abort the communicator and the process); with gloo the collective raises.  Collectives
    captured into a hipGraph are outside that mechanism (a replay creates no Work object): TrainEngine bounds those on the host with a
    ReplayGuard of the same length.  Waits that are allowed to be long (checkpoint, validation) go through ControlPlane, whose bound is
    its own.  A rank is never re-executed."""
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
        self.launched = []          # (start, end) in launch order, for tests / tracing (kept short by trim(): a run is millions of steps)
        # One entry per finished step: the buckets whose all-reduce that step issued, in order.  Ranks must agree on this sequence step
        # by step — a rank that issues a different one leaves its peers waiting in a collective (bounded: init_distributed) — whatever
        # mix of replayed graphs and plain launches each of them runs (TrainEngine keeps it for replays too).
        self.history = []
        self._step_launched = []    # this step's buckets so far

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
            self._step_launched.append((s, e))
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
        self.history.append(tuple(self._step_launched) if (self.world > 1 or self.force) else ())
        self._step_launched = []
        if len(self.history) > 2 * self.KEEP:
            self.trim()

    KEEP = 2048

    def trim(self):
        """Bound the bookkeeping lists: the newest KEEP steps / launches stay.  Indices into `history` taken before a trim are
        void; TrainEngine trims before it marks a position, so that the finish() of the capture that follows cannot."""
        if len(self.history) > self.KEEP:
            del self.history[:-self.KEEP]
        if len(self.launched) > self.KEEP:
            del self.launched[:-self.KEEP]

    def reset(self):
        """Forget a half-issued step (a hipGraph capture of it was aborted): no handles, first bucket next."""
        self._handles = []
        self._staged = []
        self._next = 0
        if self._step_launched:
            del self.launched[-len(self._step_launched):]
        self._step_launched = []

    def bucket_bytes(self):
        """Bytes of each bucket's all-reduce, in launch order."""
        return [int((e - s) * self.flat_grad.element_size()) for s, e in self.buckets]

    def grad_scale(self, grad_acc_step=1):
        return 1.0 / (grad_acc_step * self.world)
