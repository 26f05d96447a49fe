# round 5: first runs of the tall level-0 kernel (tile cfg 27): parity, then timing against cfg 23
export TMPDIR=/tmp
O=gpurun_out/r05b
mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "test_conv_pp_matches_pipe and 27" 2>&1 | tail -15 | tee $O/pytest_tall.txt
L0="8:64:64:32:1024"
{
for c in 23 27; do
  timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg $c
  timeout 120 python devtools/conv_time.py $L0 --gn --emit --cfg $c
  timeout 120 python devtools/conv_time.py $L0 --cfg $c
  timeout 120 python devtools/conv_time.py 8:64:128:32:1024 --cfg $c
  timeout 120 python devtools/conv_time.py 8:64:64:16:512 --gn --res --emit --cfg $c
done
} 2>&1 | grep -E "cfg" | tee $O/conv.txt
