"""profiles/pmc_knn.json from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE ...) of
tools/knn_time.py: the clock the chip held during knn_screen_kernel (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, MI355X_MICROARCH.md
'DVFS give-back') and how busy its matrix pipes were.   usage: pmc_knn_json.py <pmc dir> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
d, out = sys.argv[1], sys.argv[2]
dur = defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
k = "knn_screen_kernel"
c = {n: s / m for n, (m, s) in acc[k].items()}
ns = sum(dur[k]) / len(dur[k])
cyc = c["GRBM_GUI_ACTIVE"] / 8
res = {"source": "rocprofv3 --kernel-trace --pmc " + " ".join(sorted(c)) + " -- python3 tools/knn_time.py (one pair alone on the GPU)",
       "csrc_sha16": bench.csrc_digest(),
       "knn_screen_kernel_launch_us_under_pmc": ns / 1e3,
       "knn_screen_kernel_effective_clock_ghz": cyc / ns,
       "knn_screen_kernel_mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc,
       "knn_screen_kernel_mfma_instructions": c["SQ_INSTS_MFMA"],
       "knn_screen_kernel_valu_per_mfma": (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"],
       "counters_mean_per_launch": c}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({a: b for a, b in res.items() if a != "counters_mean_per_launch"}))
