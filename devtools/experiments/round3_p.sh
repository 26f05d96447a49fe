# what does the open drain of a block's last tile cost?  (variant without it: wrong results, timing only)
for lib in "" devtools/variants/liblc_nodrain.so; do
echo "== ${lib:-product}"
LC_HIP_LIB=${lib:-lidarcrafter_amd/liblidarcrafter_hip.so} python devtools/conv_time.py 8:64:64:32:1024 8:128:64:32:1024 8:64:128:32:1024 --gn --emit
LC_HIP_LIB=${lib:-lidarcrafter_amd/liblidarcrafter_hip.so} python devtools/conv_time.py 8:64:64:32:1024 --gn --emit --res
done
