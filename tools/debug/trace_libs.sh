# GPU box: per-kernel average durations (rocprofv3 kernel trace of the replayed FS2 step) under two builds of the library.
# usage: bash tools/debug/trace_libs.sh old.so new.so 'substring|substring'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/trace_libs; mkdir -p $O
i=0
for L in $1 $2; do
  export TTSK_LIB_PATH=$L
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$i -o t -- /usr/bin/python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-mel --no-e2e --no-hifi --no-extra > $O/t$i.log 2>&1
  python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('$O/t$i/t_kernel_stats.csv')))
tot=0
for r in rows:
    n=r['Name']
    if re.search(r'$3', n):
        print('$L'.split('/')[-1], n[:90].ljust(90), r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
  i=$((i+1))
done
