# round 4, call v: N consecutive runs of every test that consumes producer-side GroupNorm statistics through a model or a
# composed pipeline (goldens of all configurations, trajectories, the under-load entry check, the fused-norm op tests) --
# the statistics-entry hazard's regression surface -- plus the tests added after the last full suite.
mkdir -p gpurun_out/r04v
N=${1:-3}
SEL='golden or under_load or producer_stats or pair_stats or epilogue_stats or trajectory or batch_invariance or 64x2048 or prepare_model'
for i in $(seq 1 $N); do
  t0=$SECONDS
  timeout 400 python -m pytest tests/test_hip_parity.py tests/test_bench_shapes.py tests/test_composed_configs.py tests/test_presplit.py -m gpu -q -x -p no:cacheprovider -k "$SEL" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tr "\n" " "; echo " [run $i: $((SECONDS - t0)) s wall]"
done | tee gpurun_out/r04v/runs_$N.txt
timeout 300 python -m pytest tests/test_training.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tee -a gpurun_out/r04v/runs_$N.txt
