"""One-off: carry stored counter summaries from the byte hash of rounds 1-2 to the comment-insensitive hash of rustlight_amd/provenance.py.
Only entries whose stored hash equals the OLD hash of the tree this runs on — i.e. collected on byte-identical kernel sources — are re-stamped."""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustlight_amd import provenance
old, new = provenance.kernel_source_hash_v1(), provenance.kernel_source_hash()
def restamp(o):
    n = 0
    if isinstance(o, dict):
        if o.get("kernel_src_hash") == old: o["kernel_src_hash"] = new; o["kernel_src_hash_v1"] = old; n += 1
        for v in o.values(): n += restamp(v)
    elif isinstance(o, list):
        for v in o: n += restamp(v)
    return n
for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.json"))):
    try: o = json.load(open(p))
    except Exception: continue
    n = restamp(o)
    if n:
        json.dump(o, open(p, "w"), indent=1 if p.endswith("pmc_live.json") or "pmc_fused" in p else None)
        print(os.path.basename(p), n, old, "->", new)
