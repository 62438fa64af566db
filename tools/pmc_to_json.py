"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, one counter per pass) of one command.
HBM bytes per launch = 2 * FETCH_SIZE (gfx950 tallies the 128-B requests of wide reads at 64 B: MI355X_MICROARCH.md, HBM)
+ WRITE_SIZE, both reported in KB.   usage: pmc_to_json.py <fetch_dir> <write_dir> <out.json> "<command that was profiled>" <passes it ran> """
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from collections import defaultdict


def means(d, counter, by_grid=False):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            if by_grid:
                k = (k, int(r["Grid_Size"]))
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items()}


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
fetch_g, write_g = means(sys.argv[1], "FETCH_SIZE", True), means(sys.argv[2], "WRITE_SIZE", True)
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of: " + sys.argv[4],
       "unit_note": "KB as reported; hbm_bytes_per_launch = (2 * fetch_kb + write_kb) * 1024 (FETCH_SIZE doubled for gfx950's wide reads)",
       "passes_profiled": int(sys.argv[5]), "csrc_sha16": bench.csrc_digest(), "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    n, f = fetch.get(k, (0, 0.0)); _, w = write.get(k, (0, 0.0))
    out["kernels"][k] = {"launches_in_pass": n, "fetch_kb_per_launch": round(f, 1), "write_kb_per_launch": round(w, 1),
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
# the same per launch geometry (grid size in threads): a batched BCD launch carries several passes
for (k, grid) in sorted(set(fetch_g) | set(write_g)):
    n, f = fetch_g.get((k, grid), (0, 0.0)); _, w = write_g.get((k, grid), (0, 0.0))
    out["kernels"][k].setdefault("by_grid_threads", {})[str(grid)] = {
        "launches_in_pass": n, "fetch_kb_per_launch": round(f, 1), "write_kb_per_launch": round(w, 1),
        "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out["kernels"]), "kernels")
