# x loads two chunks ahead (second register set), -DLC_DEEP_X=1 variant vs product
for lib in lidarcrafter_amd/liblidarcrafter_hip.so devtools/variants/liblc_deep.so; do echo "== $lib"
LC_HIP_LIB=$lib python devtools/conv_time.py 8:64:64:32:1024 8:128:64:32:1024 8:64:128:32:1024 --gn --emit --res 2>&1 | grep us
LC_HIP_LIB=$lib python devtools/conv_time.py 8:64:64:32:1024 8:128:128:16:512 2>&1 | grep us
done
LC_HIP_LIB=devtools/variants/liblc_deep.so timeout 900 python -m pytest tests/test_hip_parity.py -q -x -k "conv and f16x2" 2>&1 | grep -E "passed|failed|Error" | tail -3
LC_HIP_LIB=devtools/variants/liblc_deep.so python bench.py --steps 20 --warmup 5 --repeat 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c100-420
python bench.py --steps 20 --warmup 5 --repeat 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c100-260
