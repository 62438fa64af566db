"""Per-kernel means of rocprofv3 --pmc counters (…counter_collection.csv) next to the kernel durations (…kernel_trace.csv)."""
import csv, glob, sys
from collections import defaultdict
for d in sys.argv[1:]:
    dur = defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0][:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:44]
            a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        for k in sorted(acc):
            ds = dur.get(k, [])
            print("%-44s launches %4d  mean %.1f us" % (k, len(ds), sum(ds) / max(1, len(ds))))
            for c in sorted(acc[k]):
                n, s = acc[k][c]
                print("    %-28s mean/launch %16.0f" % (c, s / n))
