#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprofv3 kernel trace.  Output -> gpurun_out/.
# usage: bash tools/gpu_check.sh [pytest-args...]   (default: the whole -m gpu suite)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
ARGS=${@:-tests}
timeout 900 python -m pytest $ARGS -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
grep -E "rel-RMS|passed|failed|Error|error" $O/pytest_gpu.log | tail -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 4000 $O/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o fs2 -- /usr/bin/python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-mel --no-e2e > $O/prof_bench.log 2>&1; echo "rocprof rc=$?"
find $O/prof -name "*stats*" | head
