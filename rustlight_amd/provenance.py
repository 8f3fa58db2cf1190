"""Provenance of stored profiler numbers: a hash of every file the device code is compiled from, comments and whitespace stripped (+ the
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


def strip_comments(text: str) -> str:
    """C / C++ source without comments and without blank or whitespace-only differences: what the compiler sees, so that rewording a
    comment does not invalidate the counters collected on that code."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":                      # string / character literal: copied verbatim
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append(" "); i = n if j < 0 else j + 2
        else:
            out.append(c); i += 1
    lines = [" ".join(l.split()) for l in "".join(out).splitlines()]
    return "\n".join(l for l in lines if l)


def kernel_source_hash() -> str:
    from . import build as rl_build

    h = hashlib.sha256()
    for p in kernel_source_files():
        h.update(os.path.relpath(p, _CSRC).encode())
        h.update(strip_comments(open(p, encoding="utf-8").read()).encode())
    h.update(" ".join(rl_build.COMMON + rl_build.HIP_EXTRA).encode())
    return h.hexdigest()[:16]


def kernel_source_hash_v1() -> str:
    """The hash of rounds 1-2 (raw file bytes, comments included); only scratch/restamp_pmc.py uses it, to carry summaries collected on
    byte-identical sources over to the comment-insensitive hash."""
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
