#!/bin/bash
# kernel stats of the kNN stage alone: bash scratch/prof_knn.sh <tag>   -> gpurun_out/<tag>_knn_stats.csv
R=$PWD; O=$R/gpurun_out/prof_$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 $R/scratch/knn_time.py $2 > $O/out.txt 2> $O/err.txt
cat $O/out.txt
cp $O/s_kernel_stats.csv $R/gpurun_out/$1_knn_stats.csv
cut -d, -f1-6 $R/gpurun_out/$1_knn_stats.csv | head -14
