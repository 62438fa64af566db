"""Per-kernel mean of a rocprofv3 --pmc counter (counter_collection.csv), KB per launch for FETCH_SIZE / WRITE_SIZE."""
import csv, glob, sys
from collections import defaultdict
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: [0, 0.0]); name = "?"
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]; name = r["Counter_Name"]
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
        for k in sorted(acc):
            print("%-12s %-40s launches %4d  mean/launch %14.0f" % (name, k, acc[k][0], acc[k][1] / acc[k][0]))
