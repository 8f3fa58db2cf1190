"""Turn the SQ / TCC counter passes of scratch/pmc_fused.sh into one small JSON (profiles/…_pmc_fused_summary.json)."""
import csv, glob, json, sys
root, kern, ms = sys.argv[1], sys.argv[2], float(sys.argv[3])
c = {}
for f in glob.glob(root + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
simd_quads = 1024 * 2.4e9 / 4 * ms * 1e-3          # VALU issue slots of the chip during the kernel (wave64 on SIMD16: 4 cycles / instr)
out = {"kernel": kern, "kernel_ms": ms, "counters": c,
       "valu_issue_busy": c.get("SQ_INSTS_VALU", 0) / simd_quads,
       "lane_utilisation": c.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64 * c.get("SQ_ACTIVE_INST_VALU", 1)),
       "waves": c.get("SQ_WAVES"),
       "l2_hit_rate": (c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])) if "TCC_HIT_sum" in c else None,
       "note": "rocprofv3 --pmc passes over bench.py --steps 1 --warmup 0 (cbox 1920x1080x128spp); clock assumed 2.4 GHz"}
print(json.dumps(out, indent=1))
