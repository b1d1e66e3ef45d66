"""In-tree build of libnphm_amd.so (hipcc, gfx950 only).  The .so is git-ignored but travels to
the GPU box with the repo snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["prep_kernels.hip", "eval_kernel.hip", "mlp_kernel.hip", "ident_bwd_kernel.hip", "mc_device.hip", "marching_cubes.cpp"]
OUT = os.path.join(HERE, "libnphm_amd.so")


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "nphm_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-fno-honor-nans",
           "-I", os.path.join(HERE, "..", "include")]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
