#!/bin/bash
out=gpurun_out/batch_exp.txt; : > $out
for rep in 1 2; do
for cfg in "8" "10" "12"; do
  line=$(python3 scratch/bench_variant.py b16.so --no-cpu-baseline --batch $cfg --steps 120 --warmup 24 | tail -1)
  echo "batch $cfg $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(round(d['ms_per_step'],3))" "$line")" | tee -a $out
done; done
