# round 4, call z: the 1x1 projections of the layout model per tile configuration (is a wider output-channel tile, which
# splits the fp32 input fewer times, ahead of the heuristic's 64-channel tiles?)
mkdir -p gpurun_out/r04z2
for cfg in 0 1 2 4 23; do
  timeout 50 python devtools/conv_time.py 8:256:768:8:256:1 8:256:256:8:256:1 8:512:1536:4:128:1 8:512:256:8:256:1 --cfg $cfg 2>&1 | grep " us"
done | tee gpurun_out/r04z2/conv1x1_cfgs.txt
