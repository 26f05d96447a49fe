S="8:64:64:32:1024:3 8:128:64:32:1024:3 8:64:128:32:1024:3 8:256:256:8:256:3"
echo "== new plain";     python devtools/conv_bench.py $S
echo "== new gn emit";   python devtools/conv_bench.py --gn --emit $S
echo "== new gn emit res"; python devtools/conv_bench.py --gn --emit --res $S
echo "== old plain";     python devtools/conv_bench.py --lib devtools/variants/liblc_prev.so $S
echo "== old gn emit res"; python devtools/conv_bench.py --lib devtools/variants/liblc_prev.so --gn --emit --res $S
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_range_safety.py tests/test_presplit.py tests/test_bench_shapes.py -m gpu -x -q 2>&1 | tail -5
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500
