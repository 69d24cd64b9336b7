#!/bin/bash
# usage: bash tools/pmc.sh <out-subdir> <script.py> ; runs rocprofv3 PMC passes (separate runs) and kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
S=$R/$2
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- /usr/bin/python3 $S > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -o p -- /usr/bin/python3 $S > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc2 -o p -- /usr/bin/python3 $S > $O/pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/pmc3 -o p -- /usr/bin/python3 $S > $O/pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc4 -o p -- /usr/bin/python3 $S > $O/pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc5 -o p -- /usr/bin/python3 $S > $O/pmc5.log 2>&1
ls $O/*
