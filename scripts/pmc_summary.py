"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection*.csv"), recursive=True)):
    acc = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            acc[(row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))].append(float(row.get("Counter_Value", 0)))
    print("==", f)
    for (k, c), v in sorted(acc.items()):
        print("%-62s %-12s n=%4d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
