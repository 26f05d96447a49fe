# round 5: step-level same-box A/B, new library against devtools/variants/liblc_bufepi0.so (predicated epilogues in the
# pre-split kernel AND the old per-value epilogue of the non-pipelined kernel)
export TMPDIR=/tmp
O=gpurun_out/r05z9
mkdir -p $O
for i in 1 2; do
  LC_HIP_LIB=devtools/variants/liblc_bufepi0.so timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/c2_old_$i.json
  timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/c2_new_$i.json
  LC_HIP_LIB=devtools/variants/liblc_bufepi0.so timeout 300 python devtools/bench_rows.py --only cond_layout_v6_32x1024 2>/dev/null > $O/c3_old_$i.json
  timeout 300 python devtools/bench_rows.py --only cond_layout_v6_32x1024 2>/dev/null > $O/c3_new_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05z9/c3_*.json")):
    d=json.load(open(f))["cond_layout_v6_32x1024"]; print(f.split('/')[-1], [r["ms_per_step"] for r in d])
for f in sorted(glob.glob("gpurun_out/r05z9/c2_*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], d["value"], d["ms_per_step"])
PY
