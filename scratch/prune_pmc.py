"""Drop the entries of profiles/pmc_live.json that were collected on other kernel sources than the tree's (bench.py then prints
`traffic: null` for those workloads until scratch/pmc_collect.sh is re-run on the GPU box); the dropped entries are kept under
profiles/pmc_stale/<hash>.json for the record."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustlight_amd import provenance  # noqa: E402

p = os.path.join(ROOT, "profiles", "pmc_live.json")
live = json.load(open(p))
h = provenance.kernel_source_hash()
stale = {k: e for k, e in live.items() if e.get("kernel_src_hash") != h}
if stale:
    os.makedirs(os.path.join(ROOT, "profiles", "pmc_stale"), exist_ok=True)
    for k, e in stale.items():
        q = os.path.join(ROOT, "profiles", "pmc_stale", f"{e.get('kernel_src_hash')}.json")
        old = json.load(open(q)) if os.path.exists(q) else {}
        old[k] = e
        json.dump(old, open(q, "w"), indent=1)
    json.dump({k: e for k, e in live.items() if k not in stale}, open(p, "w"), indent=1)
print(f"kernel source hash {h}: dropped {len(stale)} stale entries, kept {len(live) - len(stale)}")
