# round 5, first call: the two candidates written at the end of round 4 without a GPU (devtools/variants/*/README.md)
#   1. pre-split 1x1 conv, four tile shapes (measured once with shape 0: GroupNorm + qkv projection 74.9 -> 54.4 us)
#   2. 8-wave block of the f16x2 attention forward (never run; must be bit-identical to the shipped kernel)
# ~1 GPU-minute.  Afterwards: merge branch ps1x1-integration (devtools/variants/ps1x1/README.md) with the best tile shape.
mkdir -p gpurun_out/r05a
timeout 120 python devtools/variants/ps1x1/run.py 8 0 1 2 3 2>&1 | grep -E "^8:|Error|error|assert" | tee gpurun_out/r05a/ps1x1.txt
timeout 120 python devtools/variants/attn8w/run.py 2>&1 | grep -E "heads|Error|error|assert" | tee gpurun_out/r05a/attn8w.txt
# Also waiting, each a `git merge` away (compiled, never run): branch cfg26-8x32 (8 x 32 level-0 tile: then
#   python devtools/conv_time.py 8:64:64:32:1024 --gn --res --emit --cfg 26   against --cfg 0, and pytest -k "test_conv")
