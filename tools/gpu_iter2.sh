#!/bin/bash
# One build->measure iteration on the GPU box (via gpurun).  usage: bash tools/gpu_iter2.sh "<pytest args>" <tag> ["<bench flags>"]
#   selected GPU tests; bench.py with the given flags (default: the FS2 legs incl. eager / grad_acc / trainer loop / DP schedules, no
#   CPU baselines, no HiFi-GAN); a kernel trace of the train step with tools/timeline.py's summaries.  Output -> gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${2:-iter}; mkdir -p $O; cd $R; export TMPDIR=/tmp
BF=${3:---steps 50 --warmup 10 --no-cpu-baseline --no-e2e --no-mel --no-hifi}
if [ -n "$1" ]; then
  timeout 1200 python -m pytest $1 -m gpu -q -x -s > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_$TAG.log
  grep -E "passed|failed" $O/pytest_$TAG.log | tail -3
fi
timeout 900 python bench.py $BF > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"
python - <<PY
import json
try:
    r = json.loads(open("$O/bench_$TAG.json").read().strip().splitlines()[-1])
    print("ms_per_step", r["ms_per_step"], "frames/s", r["value"], "losses", r["final_losses"])
    for k in ("ms_per_step_eager", "ms_per_step_grad_acc4", "ms_per_step_train_loop"):
        print(k, r.get(k))
    d = r.get("dp_schedule_1gpu") or {}
    print("dp", d.get("schedule"), d.get("dp1_reducer_ms_per_step"), d.get("by_schedule"), d.get("error"), d.get("stderr_tail"))
    ro = r.get("roofline") or {}
    print("roofline", {k: ro.get(k) for k in ("frac", "avg_launch_us", "kernel_ms_per_step")}, (ro.get("whole_chip") or {}).get("frac"))
    h = r.get("hifi_gan")
    if h: print("hifi ms", h["ms_per_batch"], {k: round(v["ms"], 4) for k, v in h.get("stages", {}).items()})
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench_$TAG.err").read()[-3000:])
PY
cd /tmp
rm -rf $O/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o fs2 -- /usr/bin/python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-mel --no-e2e --no-hifi --no-extra > $O/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
T=$(find $O/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $T > $O/timeline_$TAG.txt 2>&1; head -50 $O/timeline_$TAG.txt
python3 $R/tools/timeline.py $T --real > $O/timeline_${TAG}_real.txt 2>&1
find $O/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
