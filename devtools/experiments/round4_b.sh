# round 4, call b: the shipped library (entry = four 32-bit stores, producer statistics ON by default):
# 1e8 entries per unit through the stress, then the full GPU suite three times
mkdir -p gpurun_out/r04b
timeout 600 python devtools/entry_stress.py --entries 1e8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04b/stress.txt
for i in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tr "\n" " "; echo
done | tee gpurun_out/r04b/suite.txt
