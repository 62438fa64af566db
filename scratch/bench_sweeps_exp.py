"""Timing experiment: the bench with a different number of sweeps (how much of a step is the chain kernel's duration worth?):
python scratch/bench_sweeps_exp.py <bcd_times> [bench args]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
n = int(sys.argv[1]); sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.BCD_TIMES = n
bench.main()
