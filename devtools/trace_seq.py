"""Kernel order of one step from a rocprofv3 --kernel-trace CSV: python devtools/trace_seq.py <kernel_trace.csv> [pattern]
Prints, for every launch whose name contains `pattern` (default: copyBuffer), the launches around it -- and the whole
sequence of the last step (between the last two launches of pstep_kernel) with durations."""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|lcconv::|void |_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:70]


names = [short(r["Kernel_Name"]) for r in rows]
ends = [i for i, n in enumerate(names) if "pstep" in n]
lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))
t0 = int(rows[lo]["Start_Timestamp"])
print(f"# last step: launches {lo}..{hi}  ({hi - lo} kernels)")
prev_end = None
for i in range(lo, hi):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:6.1f}"
    mark = " <==" if pat in names[i] else ""
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  {gap:12s} {names[i]}{mark}")
    prev_end = e
