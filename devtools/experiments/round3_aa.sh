# level-0 shapes on every pipelined tile configuration, current kernels (deferred epilogue, DMA weights)
for cfg in 23 12 22 25 15; do
python devtools/conv_time.py 8:64:64:32:1024 8:128:64:32:1024 8:64:128:32:1024 --gn --emit --res --cfg $cfg 2>&1 | grep us
done
