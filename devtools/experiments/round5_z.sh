# round 5: entries from the non-pipelined kernel's 1x1 launches; projection cells from an fp32 bracket (exact path on demand)
export TMPDIR=/tmp
O=gpurun_out/r05z10
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "conv1x1_statistics or projection or conv_1x1 or attention" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -8 | tee $O/pytest_a.txt
timeout 1500 python -m pytest tests/test_boundary.py tests/test_composed_configs.py tests/test_voxel_scatter.py tests/test_object_branch.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -5 | tee $O/pytest_b.txt
timeout 300 python devtools/bench_rows.py --only projection,points_in_boxes_mask 2>/dev/null > $O/rows_side.json
for s in 1 2; do LC_GN_TRACE=1 timeout 300 python devtools/cond_run.py 8 $s 2>&1 | grep -E "gn lookup|ok" > $O/gn_trace_c3_s$s.txt; done
for i in 1 2; do
  timeout 300 python devtools/bench_rows.py --only cond_layout_v6_32x1024 2>/dev/null > $O/c3_new_$i.json
done
timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/c2_new.json
python - <<'PY'
import json,glob
print(open("gpurun_out/r05z10/rows_side.json").read()[:1800])
for f in sorted(glob.glob("gpurun_out/r05z10/c3_*.json")):
    d=json.load(open(f))["cond_layout_v6_32x1024"]; print(f.split('/')[-1], [r["ms_per_step"] for r in d])
d=json.load(open("gpurun_out/r05z10/c2_new.json")); print(d["value"], d["ms_per_step"])
PY
