"""Step 0 of oracle/pin/README.md: the scene file rustlight's CLI will load and what the oracle expects it to produce."""
import hashlib, json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE)); sys.path.insert(0, ROOT)
from oracle import orc
from rustlight_amd import export, scenes

OUT = os.path.join(HERE, "out"); os.makedirs(OUT, exist_ok=True)
orc.build()
sd = scenes.cbox(64, 64)
sd.flip = True                                   # the Mitsuba loader's camera convention (scene_loader.rs, MTSSceneLoader)
export.write_mitsuba(sd, os.path.join(OUT, "cbox_64.xml"), "obj")
lines = []
s = orc.Rng(0)
lines += [f"u64 {s.next_u64():016x}" for _ in range(64)]
f = orc.Rng(0)
lines += [f"f32 {np.float32(f.next_f32()).view(np.uint32):08x}" for _ in range(8)]
m = orc.Rng(0)
for _ in range(4):
    seed = m.next_u64(); b = orc.Rng(seed)
    lines.append(f"blk {seed:016x} {b.next_u64():016x}")
open(os.path.join(OUT, "expected_draws.txt"), "w").write("\n".join(lines) + "\n")
alt = orc.Rng(0, 1)
open(os.path.join(OUT, "expected_draws_variant1.txt"), "w").write("\n".join(f"u64 {alt.next_u64():016x}" for _ in range(64)) + "\n")
img, st = orc.Scene(sd).render(master_seed=0, spp=4, stream_mode=0, eval_order=0)      # reference-order streams, reference recursion order
np.save(os.path.join(OUT, "expected_cbox_64x64x4_reference_order.npy"), img)
h = {n: hashlib.sha256(open(os.path.join(OUT, n), "rb").read()).hexdigest()[:16] for n in ("expected_draws.txt", "expected_cbox_64x64x4_reference_order.npy")}
json.dump(h, open(os.path.join(HERE, "expected.json"), "w"), indent=1)
print("wrote", OUT, h, "vertices", st["vertices"], "draws", st["rng_draws"])
