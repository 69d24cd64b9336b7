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
