#!/bin/bash
# usage: bash tools/prof.sh <out-subdir> <script.py> : rocprofv3 kernel trace + stats of one script
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- /usr/bin/python3 $R/$2 > $O/run.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/p_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-70s n=%5s avg %9.1f us  %5.1f%%" % (r["Name"].replace("(anonymous namespace)::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
