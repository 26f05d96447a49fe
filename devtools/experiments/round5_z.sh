# round 5: the pre-split kernel's statistics entries as one lane-spread 32-bit store each (were: 4 volatile stores + vmcnt(0))
export TMPDIR=/tmp
O=gpurun_out/r05z5
mkdir -p $O
timeout 1200 python -m pytest tests/test_presplit.py tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "presplit or resample or groupnorm or gn or unet or conv_pp or statistics or pair" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -5 | tee $O/pytest_a.txt
{ for e in 0 8 4; do timeout 120 python devtools/ps_time.py 8 --emit $e; done; } 2>&1 | grep "^ps" | tee $O/ps_emit.txt
for i in 1 2; do
  LC_GN_PRODUCER_STATS=0 timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/c2_P0_$i.json
  timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/c2_P1_$i.json
  LC_GN_PRODUCER_STATS=0 timeout 300 python devtools/bench_rows.py --only cond_layout_v6_32x1024 2>/dev/null > $O/c3_P0_$i.json
  LC_GN_QUAD_STATS=0 timeout 300 python devtools/bench_rows.py --only cond_layout_v6_32x1024 2>/dev/null > $O/c3_Q0_$i.json
  timeout 300 python devtools/bench_rows.py --only cond_layout_v6_32x1024 2>/dev/null > $O/c3_P1_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05z5/c3_*.json")):
    d=json.load(open(f))["cond_layout_v6_32x1024"]; print(f.split('/')[-1], [r["ms_per_step"] for r in d])
for f in sorted(glob.glob("gpurun_out/r05z5/c2_*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], d["value"], d["ms_per_step"])
PY
