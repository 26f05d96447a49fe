for m in 0 auto 1; do
  echo "== LC_SIDE_STREAM=$m"
  LC_SIDE_STREAM=$m python devtools/bench_rows.py --only uncond_32x1024,cond_layout_v6_32x1024 2>&1 | grep "\"batch\"\|ms_per_step"
done
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_bench_shapes.py -m gpu -x -q -k "unet or trajectory or c1 or c2 or c3 or cond or golden" 2>&1 | grep -E "passed|failed"
