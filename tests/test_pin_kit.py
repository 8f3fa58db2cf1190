"""oracle/pin/: the kit a maintainer with a Rust toolchain runs to pin the oracle to real rustlight (64 SmallRng draws + a 64x64x4
reference-order PFM).  Here: the kit regenerates its expected outputs, they carry the committed hashes, and the diff script accepts
the oracle's own outputs in rustlight's formats (and rejects a perturbed stream)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "oracle", "pin")


def test_pin_kit_round_trip(built, tmp_path):
    r = subprocess.run([sys.executable, os.path.join(PIN, "make_pin_inputs.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = json.load(open(os.path.join(PIN, "expected.json")))
    for name, h in want.items():
        assert hashlib.sha256(open(os.path.join(PIN, "out", name), "rb").read()).hexdigest()[:16] == h, name
    from rustlight_amd import api
    img = np.load(os.path.join(PIN, "out", "expected_cbox_64x64x4_reference_order.npy"))
    pfm = str(tmp_path / "ref.pfm")
    api.save_pfm(pfm, img)
    draws = os.path.join(PIN, "out", "expected_draws.txt")
    ok = subprocess.run([sys.executable, os.path.join(PIN, "diff_pin.py"), draws, pfm], capture_output=True, text=True)
    assert ok.returncode == 0 and "bit-exact" in ok.stdout, ok.stdout
    bad = str(tmp_path / "bad.txt")
    open(bad, "w").write(open(os.path.join(PIN, "out", "expected_draws_variant1.txt")).read())
    no = subprocess.run([sys.executable, os.path.join(PIN, "diff_pin.py"), bad, pfm], capture_output=True, text=True)
    assert no.returncode != 0
