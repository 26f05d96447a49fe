# round 5: tall kernel, top-row loads batched in front of the GroupNorm fold: parity, phases, timing
export TMPDIR=/tmp
O=gpurun_out/r05v
mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "test_conv_pp_matches_pipe and 27 or entries_stress" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -3 | tee $O/pytest.txt
{
timeout 100 python devtools/tall_phases.py 8:64:64:32:1024
timeout 100 python devtools/tall_phases.py 8:64:64:32:1024 --gn --res --emit
for c in 27 23; do
timeout 100 python devtools/conv_time.py 8:64:64:32:1024 --cfg $c
timeout 100 python devtools/conv_time.py 8:64:64:32:1024 --gn --res --emit --cfg $c
done
} 2>&1 | grep -E "cfg|waves|us per" | tee $O/top.txt
