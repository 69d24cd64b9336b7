#!/bin/bash
# GPU box: the FS2 step and HiFi-GAN batch times under two builds of the library, alternated on the SAME box (boxes differ by several
# per cent in clock, so only numbers from one call compare).
# usage: bash tools/ab_libs.sh tts_king_amd/libttsk_hip_old.so tts_king_amd/libttsk_hip.so [rounds] [extra bench flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq 1 $N); do
  for L in $A $B; do
    TTSK_LIB_PATH=$L timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-mel --no-extra --no-roofline --no-e2e "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
h=d.get('hifi_gan') or {}
print('$L', 'fs2 step %.4f ms' % d['ms_per_step'], ('hifi B=8 %.4f ms' % h['ms_per_batch']) if h else '')"
  done
done
