"""bench.py quotes profiler counters (roofline.traffic / roofline.pmc) only from profiles/pmc_live.json, and only when they were
collected on the kernel sources it runs (hash of csrc/kernels + flags).  This test fails when a stored entry has gone stale, i.e.
csrc/kernels changed and scratch/pmc_collect.sh was not re-run on the GPU box."""
import json
import os

from rustlight_amd import provenance

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_source_hash_is_stable_and_sensitive(tmp_path):
    h = provenance.kernel_source_hash()
    assert h == provenance.kernel_source_hash() and len(h) == 16
    assert any(p.endswith("wavefront.hip") for p in provenance.kernel_source_files())


def test_hash_ignores_comments_and_layout_but_not_code():
    code = 'int f(int x) {\n    return x * 2;   // double it\n}\nconst char* s = "// kept";\n'
    same = '/* header */\nint f(int x) {\n\n  return x * 2; // twice\n}\n\nconst char* s = "// kept";'
    other = code.replace("x * 2", "x * 3")
    assert provenance.strip_comments(code) == provenance.strip_comments(same)
    assert provenance.strip_comments(code) != provenance.strip_comments(other)
    assert '"// kept"' in provenance.strip_comments(code)


def test_stored_pmc_entries_match_the_kernel_sources():
    p = os.path.join(ROOT, "profiles", "pmc_live.json")
    if not os.path.exists(p):
        return          # nothing is quoted then: bench.py prints traffic = null
    live = json.load(open(p))
    h = provenance.kernel_source_hash()
    stale = {k: e.get("kernel_src_hash") for k, e in live.items() if e.get("kernel_src_hash") != h}
    assert not stale, f"profiles/pmc_live.json entries were collected on other kernel sources (now {h}): {stale} — re-run scratch/pmc_collect.sh"
    for k, e in live.items():
        assert e.get("commit") and e.get("counters") and e.get("kernel"), k
