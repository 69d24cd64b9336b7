#!/bin/bash
# GPU box: the FS2 step under two values of one environment switch (tts_king_amd/switches.py), alternated on the SAME box.
# usage: bash tools/ab_env.sh TTSK_SUMSQ_BYPRODUCT 0 1 [rounds] [extra bench flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
K=$1; A=$2; B=$3; N=${4:-3}; shift 4
for i in $(seq 1 $N); do
  for V in $A $B; do
    env $K=$V timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-mel --no-extra --no-roofline --no-e2e --no-hifi "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$K=$V', 'fs2 step %.4f ms' % d['ms_per_step'], 'loss %.6f' % d['final_losses']['total'])"
  done
done
