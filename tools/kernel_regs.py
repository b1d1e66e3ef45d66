"""Per-kernel register / LDS / spill figures of a hipcc -save-temps assembly file (amdhsa metadata).
usage: python tools/kernel_regs.py file.s [substring filter]"""
import re
import subprocess
import sys


def main():
    text = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    i = text.index("amdhsa.kernels:")
    blocks = re.split(r"\n  - \.agpr_count:", text[i:])[1:]
    rows = []
    for b in blocks:
        g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", b).group(1))
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        agpr = int(re.match(r"\s*(\d+)", b).group(1))
        rows.append((name, g("vgpr_count"), agpr, g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*\)$", "", n).replace("nphm::mlp::", "").replace("nphm::", "")
        if flt in n:
            print(f"{n:64s} vgpr {r[1]:4d} agpr {r[2]:3d} vspill {r[3]:3d} sspill {r[4]:3d} scratch {r[5]:5d} lds {r[6]}")


if __name__ == "__main__":
    main()
