#!/bin/bash
# GPU box: what a launch costs INSIDE the replayed train step beyond what it takes alone.  Two rocprofv3 passes with the same
# counters (kernel trace only next to --pmc, as MI355X_MICROARCH.md prescribes): the replayed FS2 step, and the flash-attention kernels alone at the
# step's shapes, plus a trace-only pass of the step; tools/chain_cost.py joins per-dispatch counters with start / end times -> gpurun_out/chain_cost/chain_cost.json.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/chain_cost; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/step -o p -- /usr/bin/python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-mel --no-e2e --no-hifi --no-extra > $O/step.log 2>&1; echo "step rc=$?"
timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/alone -o p -- /usr/bin/python3 $R/tools/debug/chain_alone.py > $O/alone.log 2>&1; echo "alone rc=$?"
# (c) the same step with the kernel trace ALONE: rocprofv3 serialises dispatches while it collects counters (each kernel of passes a / b ran with the
#     chip to itself, ~100 us apart), so what a launch takes between its real neighbours, and the gap in front of it, come from this pass
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/graph -o p -- /usr/bin/python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-mel --no-e2e --no-hifi --no-extra > $O/graph.log 2>&1; echo "graph rc=$?"
python3 $R/tools/chain_cost.py $O $O/chain_cost.json
find $O -name "*.csv" -size +30M -delete
