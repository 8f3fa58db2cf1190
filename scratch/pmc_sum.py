"""Sum rocprofv3 counter_collection.csv per (kernel, counter).  usage: pmc_sum.py file.csv [kernel-filter]"""
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:60]
    if flt not in k: continue
    acc[(k, row["Counter_Name"])] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):
    print(f"{k:62s} {c:28s} {v:18.0f}  launches {n[(k,c)]}")
