#!/bin/bash
# One build->measure iteration of the HiFi-GAN path on the GPU box (via gpurun): its GPU tests, ms per batch under graph
# replay, and the per-kernel stats of a traced run.  usage: bash tools/hifi_iter.sh [tag] [notests]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-hifi}; mkdir -p $O; cd $R; export TMPDIR=/tmp
if [ -z "$2" ]; then
  timeout 600 python -m pytest tests/test_hifigan_gpu.py -m gpu -q -x > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_$TAG.log
fi
python tools/debug/hifi_time.py 2>&1 | tail -1
bash tools/prof.sh hifi_$TAG tools/debug/hifi_prof.py
