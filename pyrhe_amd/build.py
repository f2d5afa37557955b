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
LIB_DIAG = os.path.join(CSRC, "libpyrhe_sketch_diag.so")     # -DPYRHE_DIAG: extra wrong-result switches for tools/ experiments


def library_path() -> str:
    """The product library -- or, for the measurement scripts under tools/ only, the file PYRHE_LIB names."""
    return os.environ.get("PYRHE_LIB") or LIB


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + INCLUDED + [HEADER])


def build_library(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """Compile ``libpyrhe_sketch.so`` for gfx950 if missing or older than its sources (``diag``: the diagnostic twin)."""
    if diag:
        return _compile(LIB_DIAG, ["-DPYRHE_DIAG"], verbose)
    if not force and not _stale():
        return LIB
    return _compile(LIB, [], verbose)


def _compile(out: str, extra, verbose: bool) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the gfx950 sketch engine")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"] + list(extra) + \
          ["-o", out + ".tmp"] + SOURCES
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    os.replace(out + ".tmp", out)
    return out
