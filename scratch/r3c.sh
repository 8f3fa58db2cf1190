O=gpurun_out/r3c; mkdir -p $O
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/bench_default.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
print("value", round(d["value"],1), "ms", round(d["ms_per_step"],2), "ref", d.get("reference_order_value"), "cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
for a in d["also"]: print(a["workload"], round(a["value"],1), round(a["ms_per_step"],1), a["kernel_ms"], round(a["frac"],4), a["image_crc32"], round(a["mean_vertices_per_sample"],3), a.get("context_build_s"))
PY
python bench.py --gpus 2 --width 640 --height 360 --spp 16 --steps 2 --warmup 1 --no-cpu-baseline --scaling strong > $O/bench_2rank_strong.json 2> $O/bench_2rank_strong.err; tail -2 $O/bench_2rank_strong.err
python bench.py --gpus 2 --width 640 --height 360 --spp 16 --steps 2 --warmup 1 --no-cpu-baseline --stream-mode reference > $O/bench_2rank_ref.json 2> $O/bench_2rank_ref.err; tail -2 $O/bench_2rank_ref.err
python - <<PY
import json
for n in ("bench_2rank_strong","bench_2rank_ref"):
    d=json.loads([l for l in open("$O/%s.json"%n) if l.startswith("{")][-1])
    print(n, d["scaling"], d["config"]["spp_total"], d["distributed"]["crc_match"], d["distributed"]["backend"], round(d["value"],1))
PY
rocm-smi --showcomputepartition --showmemorypartition 2>&1 | tail -12
