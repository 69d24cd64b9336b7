#!/bin/bash
# GPU box: HiFi-GAN per-stage device times under a list of environment settings.  usage: bash tools/hifi_stages.sh "VAR=a VAR=b"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
for S in $1; do
  env $(echo $S | tr ';' ' ') timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-mel --no-extra --no-roofline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['hifi_gan']
print('$S', '%.4f ms' % d['ms_per_batch'], ' '.join('%s=%.3f' % (k, v['ms']) for k, v in d['stages'].items()))"
done
