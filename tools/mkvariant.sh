#!/bin/bash
# tools/mkvariant.sh <name> <file.hip> <extra flags...>: tools/prof_build/<name>.so = libdflow.so with <file.hip> rebuilt with the flags
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../lk-s-2022-estimacija-pokreta_amd/csrc"
mkdir -p ../../tools/prof_build
extra=""
case $src in knn_mfma.hip) extra="-fno-honor-nans -fno-slp-vectorize";; bcd.hip) extra="-fno-honor-nans";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $extra "$@" -c $src -o /tmp/variant_$name.o
objs=""
for f in abi bcd knn knn_mfma knn_pca neighbour daisy post; do
  if [ "$f.hip" = "$src" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/prof_build/$name.so $objs
echo built tools/prof_build/$name.so
