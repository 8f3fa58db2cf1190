"""Sum the counters of one kernel over the rocprofv3 --pmc passes under a directory: python pmc_ab_sum.py <dir> <kernel substring>"""
import csv, glob, json, sys
root, kern = sys.argv[1], sys.argv[2]
c, n = {}, 0
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
dur = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
out = {"kernel": kern, "launches": len(dur), "mean_ms_under_profiler": sum(dur) / max(1, len(dur)), "counters": c}
w = c.get("SQ_WAVE_CYCLES")
if w: out["wave_time_shares"] = {k: c[k] / w for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA") if k in c}
if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"): out["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
print(json.dumps(out, indent=1))
