#!/bin/bash
# GPU box: end-to-end synthesis latency (1 utterance) and HiFi-GAN ms per batch under a list of environment settings.
# usage: bash tools/sweep_e2e.sh "VAR=a VAR=b;VAR2=c ..."
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
for S in $1; do
  env $(echo $S | tr ';' ' ') timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mel --no-extra --no-roofline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$S', 'e2e %.4f ms (eager %.3f)' % (d['e2e_synth']['latency_ms'], d['e2e_synth']['latency_ms_eager']), 'hifi B=8 %.4f ms' % d['hifi_gan']['ms_per_batch'])"
done
