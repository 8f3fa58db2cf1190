# round 3: fabric bytes / vector-memory instructions of the streaming kernel at 4.01 M triangles, exact vs tolerance build (BVH4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_4m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in exact fast; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$mode/p$i -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also --scene living_room --tris 4000000 --numerics $mode > $O/$mode.p$i.log 2>&1
  done
done
cd $R
python - <<PY
import csv, glob, collections, json
out = {}
for mode in ("exact", "fast"):
    c = collections.defaultdict(float); dur = []
    for f in glob.glob(f"$O/{mode}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_path_fused" in r["Kernel_Name"]: c[r["Counter_Name"]] += float(r["Counter_Value"])
    for f in glob.glob(f"$O/{mode}/p1/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_path_fused" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    out[mode] = {"kernel_ms": sum(dur) / max(1, len(dur)), "counters": dict(c), "fabric_GB_per_launch": (2.0 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 / 1e9,
                 "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)), "lane_utilisation": c.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64 * c.get("SQ_ACTIVE_INST_VALU", 0))}
json.dump(out, open("$O/pmc_living4m.json", "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters"} | {"vmem_rd": v["counters"].get("SQ_INSTS_VMEM_RD")} for k, v in out.items()}, indent=1))
PY
find $O -name '*.csv' -size +1M -delete
