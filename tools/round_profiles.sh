#!/bin/bash
# Everything the round's committed profiles come from, in one gpurun call: the full default bench line, a kernel trace + stats of
# the FS2 step and of the HiFi-GAN batch, the HBM-traffic PMC passes and the MFMA-utilisation PMC passes.
# usage (GPU box): bash tools/round_profiles.sh <tag>   -> gpurun_out/round_<tag>/...
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r}; O=$R/gpurun_out/round_$TAG; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
cp gpurun_out/bench_full.json $O/bench.json      # the long form (the line on stdout is the compact one the driver reads)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fs2_trace -o fs2 -- /usr/bin/python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-mel --no-e2e --no-hifi --no-extra > $O/fs2_trace.log 2>&1; echo "fs2 trace rc=$?"
T=$(find $O/fs2_trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $T --full > $O/step_timeline.txt 2>&1
python3 $R/tools/timeline.py $T --real > $O/step_timeline_real.txt 2>&1
find $O/fs2_trace -name "*kernel_trace.csv" -size +20M -delete
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hifi_trace -o hifi -- /usr/bin/python3 $R/tools/debug/hifi_prof.py > $O/hifi_trace.log 2>&1; echo "hifi trace rc=$?"
find $O/hifi_trace -name "*kernel_trace.csv" -size +20M -delete
cd $R
bash tools/pmc_bench.sh > $O/pmc_bench.log 2>&1; cp gpurun_out/pmc_bench/pmc_traffic.json $O/ 2>/dev/null; echo "pmc traffic rc=$?"
bash tools/pmc_mfma.sh > $O/pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma/mfma_util.json $O/ 2>/dev/null; echo "pmc mfma rc=$?"
bash tools/pmc_hifi.sh > $O/pmc_hifi.log 2>&1; cp gpurun_out/pmc_hifi/pmc_traffic.json $O/pmc_traffic_hifi.json 2>/dev/null; echo "pmc hifi rc=$?"
bash tools/chain_cost.sh > $O/chain_cost.log 2>&1; cp gpurun_out/chain_cost/chain_cost.json $O/ 2>/dev/null; echo "chain cost rc=$?"
rm -rf gpurun_out/chain_cost/step gpurun_out/chain_cost/alone gpurun_out/chain_cost/graph
rm -rf gpurun_out/pmc_hifi/fetch gpurun_out/pmc_hifi/write
rm -rf gpurun_out/pmc_bench/fetch gpurun_out/pmc_bench/write gpurun_out/pmc_mfma/fs2 gpurun_out/pmc_mfma/hifi
ls -la $O
