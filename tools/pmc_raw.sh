#!/bin/bash
# GPU box: rocprofv3 --pmc passes (kernel trace only) over a python script with a free list of counters; prints every counter per kernel symbol,
# averaged per dispatch.  usage: bash tools/pmc_raw.sh tools/debug/some_prof.py "<counters pass 1>" ["<counters pass 2>" ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; S=$1; shift; O=$R/gpurun_out/pmc_raw; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for CNT in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/p$i -o p -- /usr/bin/python3 $R/$S > $O/p$i.log 2>&1; echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, os, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r.get("Kernel_Name") or "?").replace("(anonymous namespace)::", "")
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in agg.items():
    if "Cijk" in k or "at::" in k or "pack_rb" in k: continue
    print(k[:90])
    print("   " + "  ".join("%s=%.4g" % (c.replace("SQ_", ""), v[0] / v[1]) for c, v in sorted(cs.items())))
PY
find $O -name "*.csv" -size +30M -delete
