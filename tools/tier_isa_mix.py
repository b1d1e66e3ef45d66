#!/usr/bin/env python3
"""Instruction mix of the three member bodies (three- / two- / single-term) of nphm::eval_kernel<MODE, 2> from the assembly of
`hipcc -save-temps -DNPHM_EVAL_PART=12 -c nphm_amd/csrc/eval_kernel.hip` (GPU-less):
usage tools/tier_isa_mix.py <eval_kernel-hip-amdgcn-amd-amdhsa-gfx950.s> [mangled kernel name].
The member loop dispatches once per (wavefront, member) on the tier; each body is a straight run of basic blocks (one per
chunk and per DMA guard) that contains the tier's MFMAs - found here by their first (largest) block.
Issue estimates: MFMA 32 cycles of matrix pipe, transcendental VALU 16, other VALU 4 (two wavefronts share a SIMD)."""
import collections
import re
import sys


def kind(i):
    op = i.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op in ("v_exp_f32_e32", "v_log_f32_e32", "v_rcp_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e32"): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_")): return "vmem"
    return "other"


def main():
    s = open(sys.argv[1]).read()
    name = sys.argv[2] if len(sys.argv) > 2 else "_ZN4nphm11eval_kernelILi2ELi2EEEvNS_8EvalArgsE"
    a = s.index(name + ":")
    body = s[a:s.index(".end_amdhsa_kernel", a)].split("\n")
    blocks, cur = [], []
    for ln in body:
        t = ln.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur); cur = []
        elif t and not t.startswith((".", ";", "//")):
            cur.append(t)
    blocks.append(cur)
    has = [any(kind(i) == "mfma" for i in b) for b in blocks]
    # a body starts with its largest block (L0 and the first chunk of lin1: > 360 instructions with MFMAs) and runs to the next start
    starts = [i for i, b in enumerate(blocks) if has[i] and len(b) > 360]
    if len(starts) != 3:
        print(f"warning: {len(starts)} body starts found (expected 3): {starts}")
    last = max(i for i in range(len(blocks)) if has[i]) + 1
    runs = [(st - 4, (starts[n + 1] - 4) if n + 1 < len(starts) else last) for n, st in enumerate(starts)]
    print(f"{name}: {len(blocks)} basic blocks, {sum(len(b) for b in blocks)} instructions")
    print("body (MFMAs)   VALU  of which quarter-rate  SALU   LDS  VMEM  waitcnt  s_nop  barriers  total | VALU issue / MFMA pipe cycles per visit")
    for lo, hi in sorted(runs):
        c = collections.Counter(kind(x) for b in blocks[lo:hi] for x in b)
        valu_cyc = 16 * c["trans"] + 4 * c["valu"]
        print(f"{c['mfma']:5d}        {c['valu'] + c['trans']:6d} {c['trans']:10d} {c['salu']:14d} {c['lds']:5d} {c['vmem']:5d} {c['wait']:8d} {c['nop']:6d} "
              f"{c['barrier']:9d} {sum(c.values()):6d} | {valu_cyc:6d} / {32 * c['mfma']:6d}")


if __name__ == "__main__":
    main()
