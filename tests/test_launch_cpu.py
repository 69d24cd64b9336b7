"""CPU: `bench.py --gpus N` without torch.distributed.run starts its own ranks (tts_king_amd/launch.py) — the environment
contract each rank receives, rank 0's stdout relayed alone, a failing rank failing the job, and the refusal (non-zero exit, no
`n_gpus: 1` line) when fewer than N devices are visible.  A world_size-2 gloo all-reduce runs through the launcher."""
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_RANK"] == str(r)
import torch, torch.distributed as dist
dist.init_process_group("gloo", rank=r, world_size=w)
t = torch.tensor([float(r + 1)])
dist.all_reduce(t)
dist.barrier()
dist.destroy_process_group()
if len(sys.argv) > 1 and sys.argv[1] == "fail" and r == 1:
    sys.exit(7)
print(json.dumps({"rank": r, "world": w, "sum": float(t)}))
"""


def test_spawn_two_ranks_relays_rank0_only():
    from tts_king_amd import launch
    buf = io.BytesIO()
    rc = launch.spawn_ranks(2, [sys.executable, "-c", CHILD], n_devices=2, stdout=buf, timeout=120)
    assert rc == 0
    lines = [ln for ln in buf.getvalue().decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {"rank": 0, "world": 2, "sum": 3.0}


def test_failing_rank_fails_the_job():
    from tts_king_amd import launch
    buf = io.BytesIO()
    rc = launch.spawn_ranks(2, [sys.executable, "-c", CHILD, "fail"], n_devices=2, stdout=buf, timeout=120)
    assert rc != 0


def test_too_few_devices_is_refused():
    from tts_king_amd import launch
    assert launch.spawn_ranks(8, [sys.executable, "-c", "print(1)"], n_devices=1, stdout=io.BytesIO()) == 3
    assert launch.wants_spawn(8, env={}) and not launch.wants_spawn(1, env={}) and not launch.wants_spawn(8, env={"WORLD_SIZE": "8", "RANK": "0"})


def test_bench_gpus_n_without_devices_exits_nonzero():
    """This container has no GPU: `python bench.py --gpus 2` must refuse, not print a 1-GPU line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        return                       # a multi-GPU host: the job would really run; not this test's business
    assert p.returncode != 0
    assert b"n_gpus" not in p.stdout
    assert b"HIP device" in p.stderr


HANG = r"""
import os, sys, time
sys.path.insert(0, os.environ["TTSK_ROOT"])
import torch, torch.distributed as dist
from tts_king_amd.parallel import init_distributed
rank, world, _ = init_distributed(backend="gloo", timeout_s=4)
t = torch.ones(4)
dist.all_reduce(t)
if rank == 1:
    time.sleep(120)              # this rank issues one collective fewer than its peer: the deadlock of a divergent schedule
    sys.exit(0)
dist.all_reduce(t)               # must fail within the bound, not hang
print("unreachable")
"""


def test_a_hung_collective_fails_the_job_within_the_bound():
    """every collective is bounded (parallel.init_distributed(timeout_s)); a rank left waiting by a peer that
    issued a different sequence of collectives raises, exits non-zero, and the launcher ends the other rank — no hang, no re-exec."""
    import time
    from tts_king_amd import launch
    env = dict(os.environ, TTSK_ROOT=ROOT)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    t0 = time.time()
    buf = io.BytesIO()
    rc = launch.spawn_ranks(2, [sys.executable, "-c", HANG], env=env, n_devices=2, stdout=buf, timeout=90)
    dt = time.time() - t0
    assert rc not in (0, 124), rc           # a rank failed (not: the launcher's own deadline)
    assert dt < 60, dt
    assert b"unreachable" not in buf.getvalue()


PORT_RACE = r"""
import os, sys
marker = os.environ["TTSK_MARKER"]
if os.environ["RANK"] == "0" and not os.path.exists(marker):
    open(marker, "w").write(os.environ["MASTER_PORT"])
    sys.stderr.write("RuntimeError: The server socket has failed to listen on any local network address. port: %s, useIpv6: false, code: -98, name: EADDRINUSE, message: address already in use\n" % os.environ["MASTER_PORT"])
    sys.exit(1)
import torch, torch.distributed as dist
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=r, world_size=w)
t = torch.ones(1); dist.all_reduce(t); dist.destroy_process_group()
if r == 0:
    print("port %s first %s sum %d" % (os.environ["MASTER_PORT"], open(marker).read(), int(t)))
"""


def test_taken_rendezvous_port_is_retried_once_on_another_port(tmp_path):
    """`free_port` documents a retry when the port is taken between bind-and-close and rank 0's bind."""
    from tts_king_amd import launch
    env = dict(os.environ, TTSK_MARKER=str(tmp_path / "first_attempt"))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    buf = io.BytesIO()
    rc = launch.spawn_ranks(2, [sys.executable, "-c", PORT_RACE], env=env, n_devices=2, stdout=buf, timeout=120)
    assert rc == 0, buf.getvalue()
    words = [ln for ln in buf.getvalue().decode().splitlines() if ln.startswith("port ")][0].split()      # (gloo prints a banner of its own)
    assert words[0] == "port" and words[4] == "sum" and words[5] == "2"
    assert words[1] != words[3]                 # the second attempt ran on a different port
    # a failure that is not a taken port is NOT retried
    n = tmp_path / "count"
    prog = "import os,sys\np=os.environ['TTSK_COUNT']\nopen(p,'a').write(os.environ['RANK'])\nsys.exit(5)"
    rc = launch.spawn_ranks(2, [sys.executable, "-c", prog], env=dict(env, TTSK_COUNT=str(n)), n_devices=2, stdout=io.BytesIO(), timeout=60)
    assert rc == 5 and sorted(n.read_text()) in (["0"], ["0", "1"], ["1"])


SLOW_RANK0 = r"""
import os, sys, time
sys.path.insert(0, os.environ["TTSK_ROOT"])
import torch, torch.distributed as dist
from tts_king_amd.parallel import ControlPlane, init_distributed
rank, world, _ = init_distributed(backend="gloo", timeout_s=3)           # the data-path bound: 3 s
ctrl = ControlPlane(timeout_s=120) if os.environ.get("TTSK_USE_CTRL") == "1" else None
g = torch.ones(8)
for step in range(1, 5):
    dist.all_reduce(g)                                # the step's gradient all-reduce
    if step == 2:                                     # a validation / checkpoint step
        if rank == 0:
            time.sleep(7)                             # rank 0's share takes longer than the data-path bound
        sums = ctrl.all_reduce_sums([1.0 + rank, 2.0]) if ctrl else None
        if ctrl:
            ctrl.barrier()
            assert sums == [3.0, 4.0], sums
if rank == 0:
    print("done", float(g[0]))
dist.destroy_process_group()
"""


def test_slow_rank0_at_a_validation_step_does_not_exhaust_the_collective_bound():
    """VERDICT r05 item 11 / SURVEY 8e, f-4 (reference loop: train.py:186-227): rank 0 spends longer than TTSK_DIST_TIMEOUT_S on a validation /
    checkpoint step.  With the control plane (its own gloo group, its own much larger bound) the other rank waits there — not in the next
    step's gradient all-reduce — and the job ends 0; the same job without it dies of the data-path bound (the control of this test)."""
    from tts_king_amd import launch
    base = dict(os.environ, TTSK_ROOT=ROOT)
    base.pop("WORLD_SIZE", None); base.pop("RANK", None)
    buf = io.BytesIO()
    rc = launch.spawn_ranks(2, [sys.executable, "-c", SLOW_RANK0], env=dict(base, TTSK_USE_CTRL="1"), n_devices=2, stdout=buf, timeout=120)
    assert rc == 0 and [ln for ln in buf.getvalue().decode().splitlines() if ln.startswith("done")] == ["done 16.0"], (rc, buf.getvalue())
    rc = launch.spawn_ranks(2, [sys.executable, "-c", SLOW_RANK0], env=dict(base, TTSK_USE_CTRL="0"), n_devices=2, stdout=io.BytesIO(), timeout=120)
    assert rc not in (0, 124)


def test_validation_batches_are_the_references_dealt_round_robin():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ttsk_train_main", os.path.join(ROOT, "train.py"))
    train = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(train)
    whole = train.validation_shard(37, 16)
    assert whole == [list(range(0, 16)), list(range(16, 32)), list(range(32, 37))]       # fs_two/evaluate.py:30-36: consecutive, last one short
    for world in (2, 3, 8):
        parts = [train.validation_shard(37, 16, r, world) for r in range(world)]
        assert sorted(sum(parts, []), key=lambda b: b[0]) == whole                      # same batches, each evaluated exactly once
    assert train.validation_shard(37, 16, 5, 8) == []                                   # more ranks than batches: an empty share is fine


def test_replay_guard_bounds_a_stuck_replay():
    """collectives inside a replayed hipGraph are invisible to the process group's timeout; the host bounds them."""
    import pytest
    from tts_king_amd.parallel import CollectiveTimeout, ReplayGuard

    class Ev:
        def __init__(self, done_at, clock): self.done_at, self.clock, self.recorded = done_at, clock, False
        def record(self): self.recorded = True
        def query(self): return self.clock[0] >= self.done_at

    clock = [0.0]
    def sleep(dt): clock[0] += dt
    made = []
    def make(done_after):
        def f():
            e = Ev(clock[0] + done_after, clock); made.append(e); return e
        return f
    g = ReplayGuard(timeout_s=5, depth=2, make_event=make(0.01), clock=lambda: clock[0], sleep=sleep)
    for _ in range(6):                           # a healthy run: each step is done long before it is two steps old
        g.before_step(); g.after_step(); sleep(0.02)
    assert len(g._events) == 2 and all(e.recorded for e in made)
    g.wait_all()
    assert len(g._events) == 0
    stuck = ReplayGuard(timeout_s=5, depth=2, make_event=make(1e9), clock=lambda: clock[0], sleep=sleep)
    stuck.before_step(); stuck.after_step(); stuck.before_step(); stuck.after_step()      # two steps may be in flight
    t0 = clock[0]
    with pytest.raises(CollectiveTimeout):
        stuck.before_step()                      # the third waits for the first: raises at the bound, not never
    assert 5.0 <= clock[0] - t0 < 5.2
    with pytest.raises(CollectiveTimeout):
        stuck.wait_all()
