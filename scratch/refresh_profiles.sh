#!/bin/bash
# Re-collects the tracked files of profiles/ (see profiles/README.md) into gpurun_out/refresh/: bash scratch/refresh_profiles.sh
set -e
R=$PWD; O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r02_bench.json 2> $O/bench.err
echo bench done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/st -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 16 --warmup 8 > $O/r02_bench_under_rocprof.json 2> $O/st.err
echo stats done
rocprofv3 --kernel-trace --stats -d $O/st1 -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --mode streams --inflight 1 > $O/st1.json 2> $O/st1.err
echo stats1 done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 16 --warmup 8 > $O/pf.json 2> $O/pf.err
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 16 --warmup 8 > $O/pw.json 2> $O/pw.err
echo write done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU -d $O/pm -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --mode streams --inflight 1 > $O/pm.json 2> $O/pm.err
echo mfma done
cd $R
cp $O/st/s_kernel_stats.csv $O/r02_kernel_stats.csv; cp $O/st1/s_kernel_stats.csv $O/r02_kernel_stats_single_stream.csv
python3 scratch/pmc_to_json.py $O/pf $O/pw $O/pmc_traffic.json "python3 bench.py --no-cpu-baseline --steps 16 --warmup 8"
python3 scratch/pmc_summary.py $O/pm > $O/r02_pmc_mfma.txt
rm -rf $O/st $O/st1 $O/pf $O/pw $O/pm
tail -1 $O/r02_bench.json | cut -c1-300
