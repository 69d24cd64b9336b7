#!/bin/bash
# One build->measure iteration on the GPU box (via gpurun): selected GPU tests, a short bench, a kernel trace of the train step
# with the per-step timeline summary (tools/timeline.py).  usage: bash tools/gpu_iter.sh "<pytest args>" [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${2:-iter}; mkdir -p $O; cd $R; export TMPDIR=/tmp
if [ -n "$1" ]; then
  timeout 900 python -m pytest $1 -m gpu -q -x > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_$TAG.log
  grep -E "passed|failed" $O/pytest_$TAG.log | tail -3
fi
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-e2e --no-mel --no-hifi --no-extra > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"
python - <<PY
import json
try:
    r = json.loads(open("$O/bench_$TAG.json").read().strip().splitlines()[-1])
    print("ms_per_step", r["ms_per_step"], "frames/s", r["value"], "losses", r["final_losses"])
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench_$TAG.err").read()[-2000:])
PY
cd /tmp
rm -rf $O/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o fs2 -- /usr/bin/python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-mel --no-e2e --no-hifi --no-extra > $O/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
T=$(find $O/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $T > $O/timeline_$TAG.txt 2>&1; head -45 $O/timeline_$TAG.txt
python3 $R/tools/timeline.py $T --full > $O/timeline_${TAG}_full.txt 2>&1
python3 $R/tools/timeline.py $T --real > $O/timeline_${TAG}_real.txt 2>&1
find $O/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
