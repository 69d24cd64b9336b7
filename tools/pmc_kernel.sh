#!/bin/bash
# GPU box: one rocprofv3 --pmc pass (kernel trace only, as MI355X_MICROARCH.md prescribes) over a python script, summarised per kernel:
# MFMA busy, LDS bank-conflict share, issue stalls.  usage: bash tools/pmc_kernel.sh tools/debug/some_prof.py [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${2:-k}; O=$R/gpurun_out/pmc_$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE"
timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/run -o p -- /usr/bin/python3 $R/$1 > $O/run.log 2>&1; echo "pmc rc=$?"
python3 - <<PY
import sys
sys.path.insert(0, "$R/tools")
import pmc_mfma_summary as S
rows = S.load("$O/run")
agg = {}
for e in rows:
    gui = e.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0: continue
    a = agg.setdefault(e["kernel"], dict(n=0, mfma=0., cyc=0., conf=0., lds=0., wi=0., wl=0., wa=0., wave=0.))
    a["n"] += 1; a["mfma"] += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.); a["cyc"] += gui / 8.
    a["conf"] += e.get("SQ_LDS_BANK_CONFLICT", 0.); a["lds"] += e.get("SQ_LDS_IDX_ACTIVE", 0.)
    a["wi"] += e.get("SQ_WAIT_INST_ANY", 0.); a["wl"] += e.get("SQ_WAIT_INST_LDS", 0.); a["wa"] += e.get("SQ_WAIT_ANY", 0.); a["wave"] += e.get("SQ_WAVE_CYCLES", 0.)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["cyc"]):
    if a["n"] < 2: continue
    print("%-74s n=%3d cycles %8.0f  mfma_busy %5.1f%%  lds_active/cycles*256CU %5.1f%%  lds_conflict %5.1f%%  wait_any %5.1f%%  issue_stall %5.1f%% (lds %5.1f%%)" % (
        k[:74], a["n"], a["cyc"] / a["n"], 100 * a["mfma"] / (a["cyc"] * 1024), 100 * a["lds"] / (a["cyc"] * 256), 100 * a["conf"] / max(a["lds"], 1),
        100 * a["wa"] / max(a["wave"], 1), 100 * a["wi"] / max(a["wave"], 1), 100 * a["wl"] / max(a["wave"], 1)))
PY
find $O -name "*.csv" -size +30M -delete
