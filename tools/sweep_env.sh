#!/bin/bash
# GPU box: the FS2 step's ms under a list of environment settings.  usage: bash tools/sweep_env.sh "VAR=a VAR=b ..." [bench flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
BF=${2:---steps 60 --warmup 15 --no-cpu-baseline --no-e2e --no-mel --no-hifi --no-extra --no-roofline}
for S in $1; do
  ms=$(env $(echo $S | tr ';' ' ') timeout 300 python bench.py $BF 2>/dev/null | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$S -> $ms ms"
done
