#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic counters of the train step's kernels, as MI355X_MICROARCH.md's HBM section
# prescribes — FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (no trace domains beside --kernel-trace).
# Output: gpurun_out/pmc_bench/{fetch,write}/..._counter_collection.csv ; summarise with tools/pmc_summary.py.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_bench; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="$R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-hifi --no-mel --no-e2e --no-extra"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- /usr/bin/python3 $ARGS > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- /usr/bin/python3 $ARGS > $O/write.log 2>&1; echo "write rc=$?"
find $O -name "*counter_collection.csv" | head
python3 $R/tools/pmc_summary.py $O $O/pmc_traffic.json
