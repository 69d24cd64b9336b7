#!/bin/bash
# GPU box: HiFi-GAN ms per batch (graph replay) under a list of environment settings.  usage: bash tools/sweep_hifi.sh "VAR=a VAR=b;VAR2=c ..."
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
for S in $1; do
  ms=$(env $(echo $S | tr ';' ' ') timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-mel --no-extra --no-roofline 2>/dev/null | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['hifi_gan']['ms_per_batch'])")
  echo "$S -> $ms ms"
done
