#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprofv3 kernel trace.  Output -> gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
tail -3 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 3000 $O/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o fs2 -- /usr/bin/python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/prof_bench.log 2>&1; echo "rocprof rc=$?"
ls -R $O/prof | head -30
