#!/bin/bash
# counters of the kNN kernels (one pair, kNN only): bash scratch/pmc_res.sh [variant.so]
R=$PWD
V=$1
tag=${V:-default}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_res_sq_$tag -o r --output-format csv -- python3 $R/scratch/knn_time.py $V > $R/gpurun_out/pmc_res_sq_$tag.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $R/gpurun_out/pmc_res_ta_$tag -o r --output-format csv -- python3 $R/scratch/knn_time.py $V > $R/gpurun_out/pmc_res_ta_$tag.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_res_w_$tag -o r --output-format csv -- python3 $R/scratch/knn_time.py $V > $R/gpurun_out/pmc_res_w_$tag.log 2>&1
cd $R
python3 scratch/pmc_summary.py gpurun_out/pmc_res_sq_$tag gpurun_out/pmc_res_ta_$tag gpurun_out/pmc_res_w_$tag > gpurun_out/pmc_res_$tag.txt 2>&1
grep -A9 resolve gpurun_out/pmc_res_$tag.txt
