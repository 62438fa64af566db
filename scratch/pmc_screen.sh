#!/bin/bash
# counters of knn_screen_kernel about the overlap of matrix and vector instructions: bash scratch/pmc_screen.sh
R=$PWD; O=$R/gpurun_out/pmc_screen; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $O/a -o r --output-format csv -- python3 $R/scratch/knn_time.py > $O/a.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/b -o r --output-format csv -- python3 $R/scratch/knn_time.py > $O/b.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_SALU SQ_IFETCH GRBM_GUI_ACTIVE -d $O/c -o r --output-format csv -- python3 $R/scratch/knn_time.py > $O/c.log 2>&1
cd $R
python3 scratch/pmc_summary.py $O/a $O/b $O/c > gpurun_out/pmc_screen.txt 2>&1
grep -A9 "^knn_screen" gpurun_out/pmc_screen.txt
