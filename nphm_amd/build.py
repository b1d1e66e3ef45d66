"""In-tree build of libnphm_amd.so (hipcc, gfx950 only).  The .so is git-ignored but travels to
the GPU box with the repo snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["prep_kernels.hip", "eval_kernel.hip", "mlp_kernel.hip", "mlp_bwd_kernel.hip", "ident_bwd_kernel.hip", "ident_train_kernel.hip", "fit_kernels.hip", "train_loss_kernels.hip", "mc_device.hip", "probe.hip", "marching_cubes.cpp"]
OUT = os.path.join(HERE, "libnphm_amd.so")


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "nphm_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-fno-honor-nans"]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every source for gfx950 (the translation units in parallel: eval_kernel.hip alone holds six
    instantiations of the fused kernel) and link them into one shared library."""
    if not force and not _stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    inc = ["-I", os.path.join(HERE, "..", "include")]
    objdir = tempfile.mkdtemp(prefix="nphm_amd_build_")
    try:
        jobs = []
        for src in SOURCES:
            obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
            cmd = [hipcc] + FLAGS + inc + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, obj, subprocess.Popen(cmd)))
        for cmd, _, proc in jobs:
            if proc.wait() != 0:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + [obj for _, obj, _ in jobs] + ["-o", OUT + ".tmp"]
        if verbose:
            print(" ".join(link))
        subprocess.run(link, check=True)
        os.replace(OUT + ".tmp", OUT)
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
