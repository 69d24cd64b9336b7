#!/bin/bash
# GPU box: ms per step for several --steps under a few environment settings (does the chip hold its clock over a long run?)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
for S in $1; do for N in ${2:-10 50 200}; do
  ms=$(env $(echo $S | tr ';' ' ') timeout 300 python bench.py --steps $N --warmup 10 --no-cpu-baseline --no-e2e --no-mel --no-hifi --no-extra --no-roofline 2>/dev/null | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$S steps=$N -> $ms ms"
done; done
