#!/bin/bash
# bash tools/pmc_mem.sh <scratch script> <kernel prefix> [variant.so]: HBM-side traffic counters of one kernel
R=$PWD; S=$1; K=$2; V=$3; tag=$(basename $S .py)_${V:-default}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${tag}_f -o r --output-format csv -- python3 $R/tools/$S $V > $R/gpurun_out/pmc_${tag}_f.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${tag}_wr -o r --output-format csv -- python3 $R/tools/$S $V > $R/gpurun_out/pmc_${tag}_wr.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_REQ_sum TCC_READ_sum -d $R/gpurun_out/pmc_${tag}_ea -o r --output-format csv -- python3 $R/tools/$S $V > $R/gpurun_out/pmc_${tag}_ea.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_f gpurun_out/pmc_${tag}_wr gpurun_out/pmc_${tag}_ea > gpurun_out/pmc_mem_$tag.txt 2>&1
grep -A5 "^$K" gpurun_out/pmc_mem_$tag.txt
