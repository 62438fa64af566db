#!/bin/bash
# Re-collects the tracked files of profiles/ (see profiles/README.md) into gpurun_out/refresh/: bash tools/refresh_profiles.sh [rNN]
set -e
T=${1:-r04}
R=$PWD; O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/st -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-long-run --fixed-batch 0 --steps 16 --warmup 8 > $O/${T}_bench_under_rocprof.json 2> $O/st.err
echo stats done
rocprofv3 --kernel-trace --stats -d $O/st1 -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-long-run --fixed-batch 0 --steps 6 --warmup 2 --mode streams --inflight 1 > $O/st1.json 2> $O/st1.err
echo stats1 done
# the traffic passes run the driver's geometry (20 steps = groups of 6, 7, 7; 5 warm-up steps): 25 passes, nothing else
PMCARGS="--no-cpu-baseline --no-other-configs --no-long-run --fixed-batch 0 --steps 20 --warmup 5"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python3 $R/bench.py $PMCARGS > $O/pf.json 2> $O/pf.err
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o p --output-format csv -- python3 $R/bench.py $PMCARGS > $O/pw.json 2> $O/pw.err
echo write done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU -d $O/pm -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-long-run --fixed-batch 0 --steps 6 --warmup 2 --mode streams --inflight 1 > $O/pm.json 2> $O/pm.err
echo mfma done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/pk -o p --output-format csv -- python3 $R/tools/knn_time.py > $O/pk.out 2> $O/pk.err
echo knn pmc done
cd $R
cp $O/st/s_kernel_stats.csv $O/${T}_kernel_stats.csv; cp $O/st1/s_kernel_stats.csv $O/${T}_kernel_stats_single_stream.csv
python3 tools/pmc_to_json.py $O/pf $O/pw $O/pmc_traffic.json "python3 bench.py $PMCARGS" 25
python3 tools/pmc_summary.py $O/pm > $O/${T}_pmc_mfma.txt
python3 tools/pmc_knn_json.py $O/pk $O/pmc_knn.json
rm -rf $O/st $O/st1 $O/pf $O/pw $O/pm $O/pk
# the bench line LAST, with the summaries just collected in place (its traffic / clock / stale_profiles fields read them)
cp $O/pmc_traffic.json $O/pmc_knn.json profiles/
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/bench.err
echo bench done
tail -1 $O/${T}_bench.json | cut -c1-300
