"""In-tree build of the HIP extension (``hipcc --offload-arch=gfx950``); no JIT cache."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = [os.path.join(CSRC, "sketch_engine.hip")]                 # one translation unit ...
INCLUDED = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".inc", ".h")))   # ... of these pieces
HEADER = os.path.join(os.path.dirname(_HERE), "include", "pyrhe_sketch.h")
LIB = os.path.join(CSRC, "libpyrhe_sketch.so")


def library_path() -> str:
    return LIB


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + INCLUDED + [HEADER])


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile ``libpyrhe_sketch.so`` for gfx950 if missing or older than its sources."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the gfx950 sketch engine")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           "-o", LIB + ".tmp"] + SOURCES
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    os.replace(LIB + ".tmp", LIB)
    return LIB
