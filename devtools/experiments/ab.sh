# same-box A/B of two builds of the library: bash devtools/experiments/ab.sh LIB_A LIB_B [batches...]
A=$1; Bl=$2; shift 2
for round in 1 2; do
for lib in $A $Bl; do
for b in ${@:-1 8}; do
LC_HIP_LIB=$lib python bench.py --batch $b --steps 20 --warmup 5 --repeat 5 --no-verify --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'batch', d['config']['batch_per_gpu'], d['ms_per_step'], 'ms/step')"
done; done; done
