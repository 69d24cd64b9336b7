#!/bin/bash
# Runs on the GPU box (via gpurun): MFMA-utilisation counters of the shipped FS2 train-step and HiFi-GAN kernels.
# One rocprofv3 --pmc pass per workload with --kernel-trace only (no other trace domain), as MI355X_MICROARCH.md prescribes;
# summarised by tools/pmc_mfma_summary.py into gpurun_out/pmc_mfma/mfma_util.json (copy to profiles/rNN_mfma_util.json).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_mfma; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/fs2 -o p -- /usr/bin/python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-hifi --no-mel --no-e2e --no-extra > $O/fs2.log 2>&1; echo "fs2 rc=$?"
timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/hifi -o p -- /usr/bin/python3 $R/tools/debug/hifi_prof.py > $O/hifi.log 2>&1; echo "hifi rc=$?"
python3 $R/tools/pmc_mfma_summary.py $O $O/mfma_util.json
find $O -name "*.csv" -size +30M -delete
