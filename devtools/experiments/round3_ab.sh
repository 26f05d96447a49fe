# pre-split kernel: how much of a launch is the epilogue's stores + residual loads?  (variant: wrong results)
for lib in lidarcrafter_amd/liblidarcrafter_hip.so devtools/variants/liblc_ps_nostore.so; do echo "== $lib"; LC_HIP_LIB=$lib python devtools/ps_time.py 8 2>&1 | grep "^ps"; done
