# round 4, call r: small batches -- does fusing the GroupNorm into every conv (no apply + split launch) pay at batch 1 / 2?
mkdir -p gpurun_out/r04r
for mc in 64 128 512; do
  for b in 1 2; do
    LC_FUSE_GN_MAX_CO=$mc timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --repeat 5 --no-cpu-baseline --no-roofline --no-verify 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LC_FUSE_GN_MAX_CO=$mc batch $b:', d['ms_per_step'], 'ms per step')"
  done
done | tee gpurun_out/r04r/fuse.txt
