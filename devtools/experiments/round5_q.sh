# round 5: gn_apply_split with its first loads behind the fold's entry loads; in_conv emits statistics -- groupnorm tests + bench
export TMPDIR=/tmp
O=gpurun_out/r05q
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_presplit.py tests/test_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "groupnorm or presplit or unet or c2 or golden" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -5 | tee $O/pytest.txt
timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/bench.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05q/bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['verify']['ok'], d['roofline']['time_share_per_family_ms_per_step'], d['roofline']['algorithmic_bytes_per_launch'])
PY
