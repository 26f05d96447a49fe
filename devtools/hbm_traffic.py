"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command)
into profiles/<tag>_hbm_traffic.json: HBM-side bytes per launch of the dominant kernel family.
    python devtools/hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps> <out.json>
Units are KB; FETCH_SIZE x2 on gfx950 (128-B fabric reads tallied as 64 B, calibrated on
gn_stats_kernel which reads its tensor exactly once), WRITE_SIZE x1 (MI355X_MICROARCH.md, HBM)."""
import csv
import json
import sys


def fold(path, counter):
    tot, n, allk = 0.0, 0, 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        v = float(r["Counter_Value"]) * 1024.0
        allk += v
        if "conv_f16x2" in r["Kernel_Name"] and ("Li3EEEE" in r["Kernel_Name"] or ", 3>" in r["Kernel_Name"] or "tall_kernel" in r["Kernel_Name"]):
            tot += v
            n += 1
    return tot, n, allk


fetch, nf, allf = fold(sys.argv[1], "FETCH_SIZE")
write, nw, allw = fold(sys.argv[2], "WRITE_SIZE")
steps = float(sys.argv[3])
assert nf == nw and nf > 0, (nf, nw)
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --steps 4 "
              "--warmup 2 --no-cpu-baseline --no-roofline, batch 8, f16x2 conv",
    "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE x1, units KB",
    "conv3x3_launches_per_step": nf / steps,
    "conv3x3_read_bytes_per_launch": 2.0 * fetch / nf,
    "conv3x3_write_bytes_per_launch": write / nw,
    "conv3x3_bytes_per_launch": (2.0 * fetch + write) / nf,
    "step_read_bytes": 2.0 * allf / steps,
    "step_write_bytes": allw / steps,
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
