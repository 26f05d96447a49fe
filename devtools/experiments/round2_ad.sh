for m in 0 1; do
  echo "== LC_FUSE_GN_SPLITK=$m"
  LC_FUSE_GN_SPLITK=$m python devtools/bench_rows.py --only uncond_32x1024,cond_layout_v6_32x1024 2>&1 | grep "\"batch\"\|ms_per_step"
done
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_bench_shapes.py tests/test_presplit.py tests/test_range_safety.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -3
