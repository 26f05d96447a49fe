# round 5: tall kernel as the heuristic's choice: bench A/B (LC_TALL=0/1), full GPU suite
export TMPDIR=/tmp
O=gpurun_out/r05j
mkdir -p $O
LC_TALL=0 timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/bench_tall0.json
timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/bench_tall1.json
python - <<'PY'
import json
for t in (0, 1):
    d = json.loads(open(f'gpurun_out/r05j/bench_tall{t}.json').read())
    print('LC_TALL', t, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('verify', {}).get('ok'), d.get('verify'))
PY
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest.txt
