#!/bin/bash
# Same-box comparison of group plans for the driver's flags (20 timed steps, 5 warm-up).
set -e
out=gpurun_out/groups_exp.txt
: > $out
for g in "" "7,7,6" "4,8,8" "8,8,4" "4,6,6,4" "3,7,7,3" "2,8,8,2" "4,4,4,4,4" "5,5,5,5" "2,6,6,6" "6,6,6,2"; do
  for rep in 1 2; do
    if [ -z "$g" ]; then extra=""; else extra="--groups $g"; fi
    line=$(python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $extra | tail -1)
    echo "groups=[$g] $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d['value'])" "$line")" | tee -a $out
  done
done
