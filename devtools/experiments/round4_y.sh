# round 4, call y: level-0 launch (64 -> 64 @ 32 x 1024, fused GroupNorm + residual + statistics) at batch 1 / 2 / 4 per tile
# configuration -- is the heuristic's choice (>= 256 blocks of the largest tile) right when the launch is latency-bound?
mkdir -p gpurun_out/r04y2
for b in 1 2 4; do
  for cfg in 0 13 15 25 23; do
    timeout 60 python devtools/conv_time.py $b:64:64:32:1024 --gn --res --emit --cfg $cfg 2>&1 | tail -1
  done
done | tee gpurun_out/r04y2/level0_small_batch.txt
