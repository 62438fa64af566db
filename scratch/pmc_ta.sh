#!/bin/bash
# texture-addresser / L1 counters of the kNN kernels (one pair, kNN only)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/avail.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum -d $R/gpurun_out/pmc_ta1 -o ta --output-format csv -- python3 $R/scratch/knn_time.py > $R/gpurun_out/pmc_ta1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/pmc_ta2 -o ta --output-format csv -- python3 $R/scratch/knn_time.py > $R/gpurun_out/pmc_ta2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_GATE_EN1_sum -d $R/gpurun_out/pmc_ta3 -o ta --output-format csv -- python3 $R/scratch/knn_time.py > $R/gpurun_out/pmc_ta3.log 2>&1
cd $R
python3 scratch/pmc_summary.py gpurun_out/pmc_ta1 gpurun_out/pmc_ta2 gpurun_out/pmc_ta3 > gpurun_out/pmc_ta.txt 2>&1
tail -5 gpurun_out/pmc_ta1.log gpurun_out/pmc_ta2.log gpurun_out/pmc_ta3.log
