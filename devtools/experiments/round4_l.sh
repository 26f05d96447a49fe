mkdir -p gpurun_out/r04l
timeout 300 python devtools/attn_train_time.py 2:8:32:32:2048:2048 8:8:32:32:2048:2048 8:8:64:32:2048:2061 8:16:64:32:512:525 2:8:64:32:2048:2061 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04l/attn.txt
