import os, time, torch, sys
sys.path.insert(0, os.getcwd())
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
x = torch.randn(6768, 1024); w = torch.randn(1024, 1024)
for n in (torch.get_num_threads(), 16, 8):
    torch.set_num_threads(n)
    for _ in range(3): x @ w
    t = time.perf_counter()
    for _ in range(20): x @ w
    print("threads", n, "matmul ms", (time.perf_counter() - t) / 20 * 1e3)
