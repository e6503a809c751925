"""Builds libimpnn.so (the C-ABI HIP library) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
``ionic_mpnn_amd/csrc/libimpnn.so`` is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libimpnn.so"
SOURCES = ["api.hip", "layer_kernels.hip", "loader_kernels.hip", "train_kernels.hip", "encoder_plan.hip", "encoder_fused.hip", "encoder_typed.hip", "encoder_wide.hip"]
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libimpnn.so cannot be built")
    return exe


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / "common.h", CSRC / "encoder_layout.h", CSRC / "encoder_device.h",
                                           CSRC.parent.parent / "include" / "impnn.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build_library(force=False, verbose=True, extra_flags=()):
    if not force and not needs_build():
        return LIB
    objs = []
    for src in SOURCES:
        obj = CSRC / (src + ".o")
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", str(CSRC / src),
               "-o", str(obj), *extra_flags]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(str(obj))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *objs]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB




def build_variant(out_path, src_dir=None, extra_flags=()):
    """One-shot build of a library variant (for tools/ab_bench.py) from src_dir (default: csrc)."""
    src_dir = Path(src_dir) if src_dir else CSRC
    out_path = Path(out_path)
    out_path.parent.mkdir(parents=True, exist_ok=True)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", str(out_path),
           *[str(src_dir / s) for s in SOURCES], *extra_flags]
    subprocess.run(cmd, check=True)
    return out_path


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
