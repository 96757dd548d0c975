"""Summarise the counter CSVs written by tools/pmc.sh for the trace kernel."""
import csv, glob, os, sys, collections
d = sys.argv[1]
tot = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if "trace_kernel" not in row["Kernel_Name"]:
            continue
        tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, v in tot.items():
    print("%-28s %.6g" % (k, v))
