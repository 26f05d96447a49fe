# round 4, call j: final library of the entry-store fix: 1e8 entries per unit, then full GPU suites (pass / fail lines kept)
mkdir -p gpurun_out/r04j
timeout 600 python devtools/entry_stress.py --entries 1e8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04j/stress.txt
for i in 1 2 3 4; do
  timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tr "\n" " "; echo
done | tee gpurun_out/r04j/suite.txt
