S="8:256:768:8:256:1 8:256:256:8:256:1 8:512:1536:4:128:1 8:512:512:4:128:1 8:192:64:32:1024:1 8:128:64:32:1024:1 1:128:64:32:1024:1 1:512:256:4:128:1 8:96:64:16:256:1"
for c in 0; do echo "== cfg $c"; python devtools/conv_bench.py --cfg $c $S; done
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_range_safety.py -m gpu -x -q -k "conv or attention or blocks or cond or unet" 2>&1 | tail -2
python devtools/bench_rows.py --only uncond_32x1024,cond_layout_v6_32x1024 --quick 2>&1 | grep "\"batch\"\|ms_per_step"
