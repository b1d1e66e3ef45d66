"""TEST INFRASTRUCTURE (not product): byte-compile the reference's own model files, where they lie under /root/reference,
into oracle/_ref/ so that the REFERENCE ITSELF (not a restatement) can be timed and compared on the GPU box, where
/root/reference does not exist.  No reference source enters the repo: oracle/_ref/ holds CPython bytecode only, is
git-ignored (history stays source-only) and travels with the gpurun snapshot like the built .so.

    python oracle/build_ref.py          (run by __graft_entry__.build() when /root/reference is present)

Files: src/NPHM/models/EnsembledDeepSDF.py (EnsembledLinear, EnsembledDeepSDF, sample_point_feature,
FastEnsembleDeepSDFMirrored), src/NPHM/models/deepSDF.py (DeepSDF, DeformationNetwork), src/NPHM/models/reconstruction.py
(get_logits, get_logits_backward, deform_mesh).  Loader: oracle/ref_loader.py."""
import os
import py_compile
import sys

REF = os.environ.get("NPHM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
FILES = {"EnsembledDeepSDF": "src/NPHM/models/EnsembledDeepSDF.py",
         "deepSDF": "src/NPHM/models/deepSDF.py",
         "reconstruction": "src/NPHM/models/reconstruction.py"}


def build(verbose: bool = True) -> bool:
    """-> True if the bytecode was (re)built; False when the reference checkout is absent (the GPU box: prebuilt files are used)."""
    if not os.path.isdir(os.path.join(REF, "src", "NPHM")):
        return False
    os.makedirs(OUT, exist_ok=True)
    for name, rel in FILES.items():
        src = os.path.join(REF, rel)
        dst = os.path.join(OUT, f"{name}.pyc")
        # unchecked-hash pyc: valid without the source file next to it
        py_compile.compile(src, cfile=dst, dfile=f"<reference>/{rel}", doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        if verbose:
            print(f"oracle/_ref/{name}.pyc <- {src}")
    with open(os.path.join(OUT, "PYTHON"), "w") as f:
        f.write("%d.%d\n" % sys.version_info[:2])
    return True


if __name__ == "__main__":
    if not build():
        print(f"{REF} not found: nothing built", file=sys.stderr)
        sys.exit(1)
