#!/bin/bash
# GPU box: average kernel durations of the FS2 step (rocprofv3 --kernel-trace --stats) under an environment setting, filtered by a
# pattern.  usage: bash tools/kstat.sh "VAR=a;VAR2=b" "pattern|pattern" tag
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/kstat_$3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for kv in $(echo $1 | tr ';' ' '); do export $kv; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- /usr/bin/python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-mel --no-e2e --no-hifi --no-extra > $O/run.log 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("$O/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "")
    if re.search(r"$2", n):
        print("%-60s n=%5s avg %8.2f us  total %9.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete
