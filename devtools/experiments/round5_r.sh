# round 5: side kernels -- points-in-boxes with per-box constants in LDS, projection with the pre-atomic filter: bit-exact tests + rows
export TMPDIR=/tmp
O=gpurun_out/r05r
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "points_in_boxes or projection or project or temporal or layout_condition or roiaware or pib or next_frame" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -5 | tee $O/pytest.txt
timeout 600 python devtools/bench_rows.py --only projection,points_in_boxes_mask > $O/rows.json 2> $O/rows.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05r/rows.json'))
for k, v in d.items():
    if isinstance(v, list):
        for r in v: print(k, r)
PY
