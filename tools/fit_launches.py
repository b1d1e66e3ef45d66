#!/usr/bin/env python3
"""Launch sequence of ONE replayed fitting step from a rocprofv3 kernel trace of `bench.py --workload fitting`
(`rocprofv3 --kernel-trace --output-format csv`): usage tools/fit_launches.py <..._kernel_trace.csv> [--list].
Steps are delimited by the Adam launches of the codes (or, with the optimizer inside the graph, by the step's first kernel).
Development tool."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    if len(marks) < 8:
        print("no Adam launches found")
        return
    # a step in the middle of the trace: from behind one step's last Adam launch to the next step's last Adam launch
    per_step = 2 if sum("adam_kernel" in rows[i]["Kernel_Name"] for i in marks) == len(marks) and "pair" not in rows[marks[0]]["Kernel_Name"] else 1
    k = (len(marks) // 2) // per_step * per_step
    a, b = marks[k + per_step - 1], marks[k + 2 * per_step - 1]
    seq = rows[a + 1:b + 1]
    t0 = int(rows[a]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq)
    span = int(seq[-1]["End_Timestamp"]) - t0
    print(f"launches per step {len(seq)}, kernel time {busy / 1e3:.1f} us, step {span / 1e3:.1f} us")
    if "--list" in sys.argv:
        prev = t0
        for r in seq:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - prev) / 1e3:5.1f}  {r['Kernel_Name'][:96]}")
            prev = e


if __name__ == "__main__":
    main()
