"""How many host cores this process may really use, and torch's intra-op thread count cut down to that.

A 1-GPU share of a large host shows every core of the machine (os.cpu_count() and the affinity mask say 256) and grants 16 through the
cgroup's CPU quota.  torch then starts 128 intra-op threads; every CPU-side tensor op big enough to be split (the feeder's staging
copies, the oracle in the tests) runs on 128 threads that share 16 cores — a 6768 x 1024 x 1024 matmul takes 30 ms instead of 9
(tools/debug/host_threads.py), the GPU test suite 270 s instead of 54."""
import os


def host_cores(cap=None):
    """min(affinity mask, cgroup v2 / v1 CPU quota[, cap]), at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    # the ranks of one node share the grant (torch.distributed.run / tts_king_amd.launch export LOCAL_WORLD_SIZE)
    n //= max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    if cap is not None:
        n = min(n, int(cap))
    return max(1, n)


def fit_torch_threads(cap=None):
    """Lower torch's intra-op thread count to host_cores() when it is above it (never raises it; an explicit OMP_NUM_THREADS below the
    core count is left alone).  Returns the count in force."""
    import torch
    n = host_cores(cap)
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    os.environ.setdefault("OMP_NUM_THREADS", str(n))      # processes started from here (rank launchers, workers) inherit it
    return torch.get_num_threads()
