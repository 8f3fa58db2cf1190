"""profiles/r06_kernel_resources.csv — VGPRs, spills, scratch, static LDS and waves per SIMD of EVERY kernel instantiation of the product library, read from
the code-object notes (rustlight_amd/resources.py) — is regenerated from the library this tree builds and must equal the committed table: a change that
makes a kernel spill (or stops one spilling) has to show up in the table it is judged by."""
import csv
import os

from rustlight_amd import resources

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_resource_table_matches_the_built_kernels(built):
    rows = resources.kernel_resources()
    assert len(rows) > 150 and any(r["kernel"].startswith("k_stream_spec") for r in rows) and any(r["kernel"].startswith("k_path_fused") for r in rows)
    want = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_kernel_resources.csv"))))
    got = [{k: str(v) for k, v in r.items()} for r in rows]
    assert [r["kernel"] for r in got] == [r["kernel"] for r in want], "kernel list changed: python -m rustlight_amd.resources"
    diff = [(g["kernel"], {k: (w[k], g[k]) for k in g if g[k] != w[k]}) for g, w in zip(got, want) if g != w]
    assert not diff, f"resource usage changed (committed, built): {diff[:6]} — python -m rustlight_amd.resources rewrites the table"


def test_the_hot_kernels_keep_their_budgets(built):
    """What DESIGN.md claims about the kernels the headline numbers come from."""
    rows = {(r["object"], r["kernel"]): r for r in resources.kernel_resources()}
    fused = rows[("fused_lds.hip.o", "k_path_fused<0, false, true, 1, 0, false>")]           # the diffuse Cornell box: the headline kernel
    assert fused["vgpr"] <= 128 and fused["vgpr_spill"] <= 8 and fused["max_waves_per_simd_by_vgpr"] >= 4
    fusedq = rows[("fusedq_lds.hip.o", "k_path_fused<0, false, true, 1, 0, true>")]          # its queue-fed form: the evaluation pass of reference-order streams beside the chain pass
    assert fusedq["vgpr"] <= 128 and fusedq["vgpr_spill"] <= 12 and fusedq["max_waves_per_simd_by_vgpr"] >= 4
    spec = rows[("spec_lds.hip.o", "k_stream_spec<0, false, true, 0>")]              # its chain pass in reference-order streams
    # three waves per SIMD since the parked state was cut to 30 words: three workgroups per CU is what LDS allows, so the registers of a fourth wave buy nothing
    assert spec["vgpr"] <= 168 and spec["vgpr_spill"] == 0 and spec["max_waves_per_simd_by_vgpr"] >= 3
