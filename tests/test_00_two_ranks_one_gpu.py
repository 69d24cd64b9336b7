"""GPU: two REAL rank processes on the one MI355X of the box, started by the product launcher (tts_king_amd/launch.py) — the only
cross-process evidence a 1-GPU box allows (reference semantics: train.py:43-54, N ranks = grad_acc over N batches).

RCCL refuses two ranks per device, so the ranks talk over gloo and the GradReducer stages its buckets through pinned host memory
(`host_staged`); everything else is the product path: TrainEngine (shape buckets, hipGraph replay of the accumulate-only micro-steps,
eager update steps), _GroupNotifier's bucket announcements from backward, clip + Adam replicated on every rank.  The two ranks see
batches of different shapes that recur at different steps, so one replays a captured graph while the other still launches eagerly.

Runs FIRST in the session (file name): the parent must start its children before this process has touched the GPU."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_processes_share_one_gpu(tmp_path):
    from tts_king_amd import launch
    n_updates = 6
    worker = os.path.join(ROOT, "tests", "dp_rank_worker.py")
    with open(os.path.join(str(tmp_path), "rank0.log"), "wb") as log:
        rc = launch.spawn_ranks(2, [sys.executable, worker, str(tmp_path), str(n_updates)], n_devices=2, timeout=900, stdout=log)
    assert rc == 0, "rank processes failed or hung (rc %d; 124 = no result within the bounded wait)" % rc
    with open(os.path.join(str(tmp_path), "verdict.json")) as f:
        v = json.load(f)
    print(v)
    assert v["updates"] == [n_updates, n_updates]
    assert v["ranks_equal"], "the two ranks ended with different weights"
    assert v["equals_one_process"], "two-process result differs from the one-process accumulation: max abs %.3e" % v["max_abs_vs_one_process"]
    for st in v["stats"]:            # both ranks did capture and replay accumulate-only micro-steps, and ran their update steps eagerly
        assert st["replayed"] >= 1 and st["captured"] >= 1 and st["eager"] >= n_updates, st
