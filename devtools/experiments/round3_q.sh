# after the small-batch changes (split-apply grid, conv prologue): tests + batch 1/2/8 step times
timeout 900 python -m pytest tests -m gpu -q -x -k "conv or groupnorm or gn or norm or unet or sampl" 2>&1 | tail -3
for b in 1 2 8; do python bench.py --batch $b --steps 20 --warmup 5 --repeat 5 --no-verify --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', d['config']['batch_per_gpu'], d['ms_per_step'], 'ms/step')"; done
