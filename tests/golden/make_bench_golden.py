"""Golden CRCs of the frames bench.py times: the PARITY build of the CPU oracle renders each timed frame — same scene,
resolution, spp, stream mode and master seed as the last timed step — and the CRC-32 of the float32 image goes into bench_crcs.json.
bench.py only LOOKS THE CRC UP (no oracle import in the timed program) and prints `oracle_crc_match`; tests/test_bench_host.py checks the file,
a `-m gpu` test renders one of the frames and compares.

Run in the build container (no GPU needed; the whole set takes ~25 min on 8 cores):   python tests/golden/make_bench_golden.py [workload ...]
Key: "<workload>:<width>x<height>x<spp>:<stream mode>:seed<master seed>"; the master seed of bench.py's step s is s, so the last of K timed steps is K-1
(the driver runs --steps 20: seed 19; the default run and the `also` records: 3 steps, seed 2)."""
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc                      # noqa: E402
from rustlight_amd import scenes            # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_crcs.json")
FRAMES = [  # (workload, scene factory, width, height, spp, stream mode name, oracle stream_mode, seeds)
    ("cbox", lambda w, h: scenes.cbox(w, h), 256, 256, 16, "reference", 0, (0,)),                       # BASELINE configs[0]
    ("cbox", lambda w, h: scenes.cbox(w, h), 1920, 1080, 128, "per_sample", 1, (2, 19)),               # configs[1], the headline
    ("cbox", lambda w, h: scenes.cbox(w, h), 1920, 1080, 128, "reference", 0, (2,)),                   # configs[1] in rustlight's own streams
    ("cbox", lambda w, h: scenes.cbox(w, h), 1080, 1080, 128, "per_sample", 1, (2,)),                  # square frame
    ("living_room", lambda w, h: scenes.living_room(w, h), 1920, 1080, 128, "per_sample", 1, (2,)),    # configs[2] stand-in
    ("cbox_medium", lambda w, h: scenes.cbox_medium(w, h, 0.5), 1920, 1080, 128, "per_sample", 1, (2,)),   # configs[4]
    # round 5: the drop-in default (reference-order streams) on the two slow configs, one timed step (seed 0), and configs[3]'s per-rank workload
    # (shard 0 of 8 at 1024 spp: key suffix ":shard0of8"), both stream modes
    # (round 6: three timed steps each — the last one is seed 2)
    ("living_room", lambda w, h: scenes.living_room(w, h), 1920, 1080, 128, "reference", 0, (0, 2)),
    ("cbox_medium", lambda w, h: scenes.cbox_medium(w, h, 0.5), 1920, 1080, 128, "reference", 0, (0, 2)),
    ("cbox", lambda w, h: scenes.cbox(w, h), 1920, 1080, 1024, "reference", 0, (0,), (0, 8)),
    ("cbox", lambda w, h: scenes.cbox(w, h), 1920, 1080, 1024, "per_sample", 1, (0,), (0, 8)),
    # round 6: configs[3] WHOLE — the full 1920 x 1080 x 1024 spp frame (what the eight shards sum to), both stream modes (~7 min each on 8 cores)
    ("cbox", lambda w, h: scenes.cbox(w, h), 1920, 1080, 1024, "reference", 0, (0,)),
    ("cbox", lambda w, h: scenes.cbox(w, h), 1920, 1080, 1024, "per_sample", 1, (0,)),
]


def main():
    only = set(sys.argv[1:])
    table = json.load(open(OUT)) if os.path.exists(OUT) else {}
    threads = len(os.sched_getaffinity(0))
    for name, make, w, h, spp, mode, smode, seeds, *rest in FRAMES:
        shard = rest[0] if rest else (0, 1)
        if only and name not in only and f"{name}:{mode}" not in only:
            continue
        sc = None
        for seed in seeds:
            key = f"{name}:{w}x{h}x{spp}:{mode}:seed{seed}" + (f":shard{shard[0]}of{shard[1]}" if shard[1] > 1 else "")
            if key in table:
                continue
            if sc is None:
                sc = orc.Scene(make(w, h))
            t0 = time.time()
            img, st = sc.render(master_seed=seed, spp=spp, stream_mode=smode, eval_order=1, threads=threads, shard_index=shard[0], shard_count=shard[1])      # eval_order 1: the forward accumulation the kernels use
            table[key] = {"crc32": f"{zlib.crc32(img.tobytes()):08x}", "image_mean": float(img.mean()), "camera_samples": st["camera_samples"], "vertices": st["vertices"],
                          "rng_draws": st["rng_draws"], "oracle_seconds": round(time.time() - t0, 1)}
            print(key, table[key], flush=True)
            json.dump(table, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
