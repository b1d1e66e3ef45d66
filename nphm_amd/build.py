"""In-tree build of libnphm_amd.so (hipcc, gfx950 only).  The .so is git-ignored but travels to
the GPU box with the repo snapshot."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# (source, extra flags, object tag): eval_kernel.hip is compiled in four parts (its nine kernel instantiations are 3 of the 3.5
# minutes a one-piece build takes): the C ABI + dispatch, and the kernels of each precision (csrc/eval_kernel.hip, NPHM_EVAL_PART)
EVAL_PARTS = [("eval_kernel.hip", ["-DNPHM_EVAL_PART=1"], "eval_kernel_api"), ("eval_kernel.hip", ["-DNPHM_EVAL_PART=10"], "eval_kernel_f32"),
              ("eval_kernel.hip", ["-DNPHM_EVAL_PART=11"], "eval_kernel_bf16"), ("eval_kernel.hip", ["-DNPHM_EVAL_PART=12"], "eval_kernel_f16")]
SOURCES = ["prep_kernels.hip", "mlp_kernel.hip", "mlp_bwd_kernel.hip", "ident_bwd_kernel.hip", "ident_train_kernel.hip", "fit_kernels.hip", "train_loss_kernels.hip", "dense_train_kernels.hip", "mc_device.hip", "probe.hip", "marching_cubes.cpp"]
OUT = os.path.join(HERE, "libnphm_amd.so")
OBJ_CACHE = os.path.join(HERE, "..", ".build_cache")      # objects by content hash (git- and gpurun-ignored)


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "nphm_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-fno-honor-nans"]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every source for gfx950 (the translation units in parallel: eval_kernel.hip alone holds six
    instantiations of the fused kernel) and link them into one shared library.  Objects are kept by content hash of
    (flags, headers, source) under .build_cache/: an edit recompiles its translation unit only; ``force`` recompiles all."""
    if not force and not _stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    inc = ["-I", os.path.join(HERE, "..", "include")]
    os.makedirs(OBJ_CACHE, exist_ok=True)
    # every translation unit sees every header: one digest of them + the flags, then one of the source per object
    common = hashlib.sha256(" ".join(FLAGS).encode())
    try:                                                    # a toolchain upgrade invalidates every cached object
        common.update(subprocess.run([hipcc, "--version"], capture_output=True, check=True).stdout)
    except Exception:                                       # noqa: BLE001
        pass
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            common.update(open(os.path.join(CSRC, f), "rb").read())
    common.update(open(os.path.join(HERE, "..", "include", "nphm_amd.h"), "rb").read())
    jobs, objs = [], []
    units = EVAL_PARTS + [(src, [], os.path.splitext(src)[0]) for src in SOURCES]       # (the slowest first)
    for src, extra, tag in units:
        h = common.copy()
        h.update(" ".join(extra).encode())
        h.update(open(os.path.join(CSRC, src), "rb").read())
        obj = os.path.join(OBJ_CACHE, f"{tag}-{h.hexdigest()[:20]}.o")
        objs.append(obj)
        if os.path.exists(obj) and not force:
            continue
        tmp = f"{obj}.{os.getpid()}.tmp"                    # (two builds at once - ranks, xdist workers - never share a temp file)
        cmd = [hipcc] + FLAGS + extra + inc + ["-c", os.path.join(CSRC, src), "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        jobs.append((cmd, obj, tmp, subprocess.Popen(cmd)))
    for cmd, obj, tmp, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        os.replace(tmp, obj)
    out_tmp = f"{OUT}.{os.getpid()}.tmp"
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", out_tmp]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True)
    os.replace(out_tmp, OUT)
    # objects of older versions of OUR sources only (another build's in-flight temp files are not ours to delete)
    keep = {os.path.basename(o) for o in objs}
    stems = {tag for _, _, tag in units} | {"eval_kernel"}
    for f in os.listdir(OBJ_CACHE):
        if f.endswith(".o") and f not in keep and f.rsplit("-", 1)[0] in stems:
            os.remove(os.path.join(OBJ_CACHE, f))
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
