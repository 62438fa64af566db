#!/bin/bash
# same-box sweep of the bench's stream / group parameters
out=gpurun_out/front_exp.txt; : > $out
for rep in 1 2; do
for cfg in "--front 3 --batch 8" "--front 2 --batch 8" "--front 4 --batch 8" "--front 1 --batch 8" "--front 3 --batch 6" "--front 3 --batch 4"; do
  line=$(python3 bench.py --no-cpu-baseline $cfg | tail -1)
  echo "$cfg $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(round(d['ms_per_step'],3))" "$line")" | tee -a $out
done; done
