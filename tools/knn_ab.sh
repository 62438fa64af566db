#!/bin/bash
# same-box A/B of kNN variants: bash tools/knn_ab.sh a.so b.so ... (two rounds)
for rep in 1 2; do for v in "$@"; do python3 tools/knn_time.py $v; done; done
