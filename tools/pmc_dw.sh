#!/bin/bash
# GPU box: PMC passes over tools/debug/dw_micro.py (the weight-gradient group alone): L2 hit / miss, HBM bytes, LDS and MFMA activity.
# usage: bash tools/pmc_dw.sh [which] -> gpurun_out/pmc_dw/*.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_dw; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
W=${1:-all}
python3 $R/tools/debug/dw_micro.py $W 20 | tee $O/time_$W.txt
python3 $R/tools/debug/dw_micro.py dec 20 | tee -a $O/time_$W.txt
python3 $R/tools/debug/dw_micro.py w1 20 | tee -a $O/time_$W.txt
python3 $R/tools/debug/dw_micro.py pn 20 | tee -a $O/time_$W.txt
i=0
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o p -- /usr/bin/python3 $R/tools/debug/dw_micro.py $W 3 > $O/p$i.log 2>&1; echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k: continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/summary_$W.txt", "w") as out:
    for k, cs in acc.items():
        out.write(k + "\n")
        for c, v in sorted(cs.items()):
            out.write("   %-34s n=%3d  mean %.4g\n" % (c, len(v), sum(v) / len(v)))
print(open("$O/summary_$W.txt").read())
PY
find $O -name "*.csv" -size +5M -delete
