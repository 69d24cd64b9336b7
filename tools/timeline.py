#!/usr/bin/env python3
"""Per-dispatch timeline of ONE replayed FS2 train step from a rocprofv3 --kernel-trace CSV.

usage: python tools/timeline.py <..._kernel_trace.csv> [--step N] [--full] [--real]
A step is delimited by consecutive `adam_kernel` dispatches (the last kernel of a step).  Prints the kernels of step N
(default: the last complete one) grouped by symbol with launch counts, summed duration, and the idle gaps between
dispatches; --full lists every dispatch in order (name, grid, duration, gap before it) on a cumulative clock (gap + duration:
overlapped kernels are laid end to end); --real lists them with their TRUE start and end times relative to the step's first
kernel and the queue they ran on, so that concurrency between the step's streams — and the time a stream sat waiting for
another — can be read off."""
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name[:64]


def main():
    path = sys.argv[1]
    full = "--full" in sys.argv
    want = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else None
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("adam_kernel", "adam_clip_kernel", "adam_pack_kernel"))]
    if len(ends) < 2:
        raise SystemExit("fewer than two adam_kernel dispatches in the trace")
    n = len(ends) - 1 if want is None else want
    a, b = ends[n - 1] + 1, ends[n] + 1
    step = rows[a:b]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    prev_end = int(rows[a - 1]["End_Timestamp"])
    agg, gaps, busy = {}, 0, 0
    seq = []
    last = prev_end
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = max(0, s - last)
        d = agg.setdefault(short(r["Kernel_Name"]), [0, 0, 0])
        d[0] += 1; d[1] += e - s; d[2] += gap
        gaps += gap; busy += e - s
        seq.append((short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), e - s, gap))
        last = max(last, e)
    print("step %d: %d dispatches, wall %.1f us (first start -> last end), sum of kernel durations %.1f us, idle gaps %.1f us"
          % (n, len(step), (t1 - t0) / 1e3, busy / 1e3, gaps / 1e3))
    print("%-64s %5s %10s %8s %9s" % ("kernel", "n", "sum us", "avg us", "gaps us"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-64s %5d %10.1f %8.1f %9.1f" % (k, v[0], v[1] / 1e3, v[1] / v[0] / 1e3, v[2] / 1e3))
    if "--real" in sys.argv:
        print()
        queues = {}
        for r in step:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            q = queues.setdefault(r["Queue_Id"], len(queues))
            print("%9.1f -> %9.1f  q%d  %-56s grid %6d x%3d x%3d  %8.1f us" % ((s - t0) / 1e3, (e - t0) / 1e3, q, short(r["Kernel_Name"]),
                  int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), (e - s) / 1e3))
    elif full:
        print()
        t = 0.0
        for name, gx, gy, gz, dur, gap in seq:
            t += gap / 1e3
            print("%9.1f  %-56s grid %6d x%3d x%3d  %8.1f us  (gap %5.1f)" % (t, name, gx, gy, gz, dur / 1e3, gap / 1e3))
            t += dur / 1e3


if __name__ == "__main__":
    main()
