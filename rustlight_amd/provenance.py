"""Provenance of stored profiler numbers: a hash of every file the device code is compiled from (+ the
compiler flags).  `profiles/*pmc*.json` carry the hash of the tree they were collected on; bench.py only
quotes them when it matches the tree it runs from, and tests/test_provenance.py fails when a stored
summary that bench.py reads has gone stale."""
from __future__ import annotations

import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")


def kernel_source_files():
    out = [os.path.join(_CSRC, "device_types.h"), os.path.join(_CSRC, "detmath_shared.h")]
    kdir = os.path.join(_CSRC, "kernels")
    out += sorted(os.path.join(kdir, f) for f in os.listdir(kdir) if f.endswith((".hip", ".h")))
    return out


def kernel_source_hash() -> str:
    from . import build as rl_build

    h = hashlib.sha256()
    for p in kernel_source_files():
        h.update(os.path.relpath(p, _CSRC).encode())
        h.update(open(p, "rb").read())
    h.update(" ".join(rl_build.COMMON + rl_build.HIP_EXTRA).encode())
    return h.hexdigest()[:16]


def git_head() -> str | None:
    try:
        return subprocess.check_output(["git", "-C", os.path.dirname(_HERE), "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:
        return None
