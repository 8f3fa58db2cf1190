"""Turn the counter passes of scratch/pmc_collect.sh into one JSON (stdout), stamped with the hash of the kernel sources and the commit
they were collected on.   usage: pmc_summary.py <dir> <kernel-filter> [bench.py args...]
    pmc_summary.py --merge <summary.json>      merge a summary into profiles/pmc_live.json (what bench.py may quote)
The VALU issue model comes from profiles/r02_valu_calibration.json (scratch/valu_calib.hip), not from an assumption."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustlight_amd import provenance

def calibration():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r02_valu_calibration.json")))
    except Exception:
        return {}

def main():
    if sys.argv[1] == "--merge":
        e = json.load(open(sys.argv[2]))
        p = os.path.join(ROOT, "profiles", "pmc_live.json")
        live = json.load(open(p)) if os.path.exists(p) else {}
        live[e["key"]] = e
        json.dump(live, open(p, "w"), indent=1)
        print("merged", e["key"], "->", p)
        return
    root, kern, bargs = sys.argv[1], sys.argv[2], sys.argv[3:]
    import bench
    a = bench.parse_args(bargs)
    c, dur, n_launch = {}, [], 0
    for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for f in glob.glob(root + "/p1/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    ms = sum(dur) / max(1, len(dur))
    cal = calibration()
    cyc_valu = cal.get("cycles_per_wave64_valu_instruction", None)
    n_simd = 256 * 4
    clk_ghz = min(2.4, c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / (ms * 1e-3) / 1e9) if c.get("GRBM_GUI_ACTIVE") and ms else None     # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    out = {"key": f"{a.scene}{':tris' + str(a.tris) if a.tris else ''}:{a.width}x{a.height}x{a.spp}:{a.stream_mode}:{a.numerics}", "kernel": kern, "kernel_ms_under_profiler": ms, "launches_seen": len(dur),
           "kernel_src_hash": provenance.kernel_source_hash(), "commit": os.environ.get("RL_COMMIT"), "bench_args": bargs, "counters": c,
           "effective_clock_GHz": clk_ghz}
    if ms and cyc_valu and "SQ_INSTS_VALU" in c:
        clk = (clk_ghz or 2.4) * 1e9
        out["valu_issue_model"] = {"cycles_per_wave64_valu_instruction": cyc_valu, "source": "profiles/r02_valu_calibration.json",
                                   "valu_instructions": c["SQ_INSTS_VALU"], "issue_slots_available": n_simd * clk * ms * 1e-3 / cyc_valu,
                                   "valu_issue_busy": c["SQ_INSTS_VALU"] * cyc_valu / (n_simd * clk * ms * 1e-3)}
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        out["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        out["wave_time_shares"] = {k: c.get(k, 0.0) / c["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")}
    if "TCC_HIT_sum" in c:
        out["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0))
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        # rocprofv3 reports KB; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE as reported
        out["hbm_bytes_per_launch"] = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0 / max(1, len(dur))
        out["hbm_note"] = "FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes), per launch; counters collected in their own rocprofv3 --pmc passes"
    out["waves"] = c.get("SQ_WAVES")
    out["bound"] = "valu"
    print(json.dumps(out, indent=1))

main()
