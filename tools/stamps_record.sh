#!/bin/bash
# GPU box: the phase stamps (s_memrealtime inside the kernels, diagnostic build `make -C tts_king_amd/csrc stamps`) of the decoder's
# LayerNorm-family and window-conv kernels and of the C = 128 pair kernel, as text records for profiles/ ("a stamp record
# that shows which phase paid").  usage: bash tools/stamps_record.sh [out dir]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=${1:-gpurun_out/stamps}; mkdir -p $O
export TTSK_LIB_PATH=tts_king_amd/libttsk_hip_stamps.so
for t in lnb_stamps wl_stamps wc_stamps pair_stamps; do
  timeout 300 python tools/debug/$t.py > $O/$t.txt 2>&1; echo "$t rc=$?"
done
