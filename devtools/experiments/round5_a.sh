# round 5, first call: the queue left by round 4 (cfg 26 = 8 x 32 tile, integrated pre-split 1x1, 8-wave attention),
# the box's baseline, and the level-0 ablations that size the round's kernel work (x loads / stores+residual / staging
# arithmetic / MFMAs compiled out: wrong results, timing only).
export TMPDIR=/tmp
O=gpurun_out/r05a
mkdir -p $O
L0="8:64:64:32:1024"
{
echo "== conv_time level-0 64->64 full (gn res emit), cfg 0/23/26/226/426"
for c in 0 23 26 226 426; do timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg $c; done
echo "== plain"
for c in 23 26; do timeout 120 python devtools/conv_time.py $L0 --cfg $c; done
echo "== 128->64 full, 64->128 plain"
for c in 23 26; do timeout 120 python devtools/conv_time.py 8:128:64:32:1024 --gn --res --emit --cfg $c; timeout 120 python devtools/conv_time.py 8:64:128:32:1024 --cfg $c; done
echo "== ablations of cfg 23 (full): abl2 no x loads, abl24 no stores / residual loads, abl1 no staging VALU + ds_write, abl32 no MFMA, aux0 write-back deferred stores"
for v in abl2 abl24 abl1 abl32 aux0; do
  echo "-- $v"; LC_HIP_LIB=devtools/variants/liblc_$v.so timeout 120 python devtools/conv_time.py $L0 --gn --res --emit --cfg 23
done
} 2>&1 | grep -vE "^$|Warning|warn" | tee $O/conv.txt
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "test_conv and (26 or 226)" 2>&1 | tail -3 | tee $O/pytest_cfg26.txt
timeout 300 python -m pytest tests/test_presplit.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_ps1x1.txt
timeout 120 python devtools/variants/ps1x1/run.py 8 0 1 2 3 2>&1 | grep -E "^8:|Error|error|assert" | tee $O/ps1x1.txt
timeout 120 python devtools/variants/attn8w/run.py 2>&1 | grep -E "heads|Error|error|assert" | tee $O/attn8w.txt
timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 | tee $O/bench.json | head -c 600
