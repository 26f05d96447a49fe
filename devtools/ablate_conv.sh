#!/bin/bash
# Developer ablation of the pipelined f16x2 conv (run on the GPU box): rebuild the library with
# -DLC_ABLATE=<mask> and time one shape.  1: no split/ds_write, 2: no global loads, 4: no ds_reads.
for m in 0 1 2 3 4 7; do
  LC_EXTRA_HIPCC_FLAGS="-DLC_ABLATE=$m" python -m lidarcrafter_amd.build --force > /dev/null 2>&1
  echo "ablate=$m"
  python devtools/conv_time.py 8 256 256 8 256 3 23
  python devtools/conv_time.py 8 256 256 8 256 3 12
  python devtools/conv_time.py 8 64 64 32 1024 3 23
done
python -m lidarcrafter_amd.build --force > /dev/null 2>&1
