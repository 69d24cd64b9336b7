#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic counters of the HiFi-GAN generator's kernels (tools/debug/hifi_prof.py), FETCH_SIZE
# and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes.  Output: gpurun_out/pmc_hifi/pmc_traffic.json (tools/pmc_summary.py).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_hifi; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="$R/tools/debug/hifi_prof.py"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- /usr/bin/python3 $ARGS > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- /usr/bin/python3 $ARGS > $O/write.log 2>&1; echo "write rc=$?"
python3 $R/tools/pmc_summary.py $O $O/pmc_traffic.json
