#!/bin/bash
R=$PWD; O=$R/gpurun_out/chain_phases; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python3 $R/scratch/chain_phases.py > $O/out.txt 2>&1
cd $R
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/chain_phases/t_kernel_trace.csv')) if 'bcd_chain' in r['Kernel_Name']]
for r in rows[-28:]:
    print(r['Grid_Size_X'], r['Grid_Size_Y'], round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1), 'us')
PY
