"""bench.py against an alternative libdflow.so (tools/prof_build/<name>): python tools/bench_variant.py <lib> [bench args]"""
import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
_lib = importlib.import_module("lk-s-2022-estimacija-pokreta_amd._lib")
_lib.LIB_PATH = os.path.join(ROOT, "tools", "prof_build", sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
