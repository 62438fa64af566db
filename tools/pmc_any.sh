#!/bin/bash
# bash tools/pmc_any.sh <scratch script> <kernel name prefix> [variant.so]: SQ / TA / wait counters of one kernel
R=$PWD; S=$1; K=$2; V=$3; tag=$(basename $S .py)_${V:-default}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${tag}_sq -o r --output-format csv -- python3 $R/tools/$S $V > $R/gpurun_out/pmc_${tag}_sq.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $R/gpurun_out/pmc_${tag}_ta -o r --output-format csv -- python3 $R/tools/$S $V > $R/gpurun_out/pmc_${tag}_ta.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${tag}_w -o r --output-format csv -- python3 $R/tools/$S $V > $R/gpurun_out/pmc_${tag}_w.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum -d $R/gpurun_out/pmc_${tag}_l2 -o r --output-format csv -- python3 $R/tools/$S $V > $R/gpurun_out/pmc_${tag}_l2.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_sq gpurun_out/pmc_${tag}_ta gpurun_out/pmc_${tag}_w gpurun_out/pmc_${tag}_l2 > gpurun_out/pmc_$tag.txt 2>&1
grep -A9 "^$K" gpurun_out/pmc_$tag.txt
