# round 4, call a: do the 32-bit / no-soffset entry-store variants cure the statistics-entry corruption?
# control = the shipped 128-bit store with SGPR soffset.  (profiles/r04_entry_store.txt)
mkdir -p gpurun_out/r04a
for lib in lidarcrafter_amd/liblidarcrafter_hip.so devtools/variants/liblc_entry_m1.so devtools/variants/liblc_entry_m2.so; do
  LC_HIP_LIB=$lib LC_GN_PRODUCER_STATS=1 timeout 400 python devtools/entry_stress.py --entries 5e7 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04a/stress.txt
for lib in lidarcrafter_amd/liblidarcrafter_hip.so devtools/variants/liblc_entry_m1.so devtools/variants/liblc_entry_m2.so; do
  LC_HIP_LIB=$lib python devtools/conv_time.py 8:64:64:32:1024 --gn --emit --res 2>&1 | grep us
done | tee gpurun_out/r04a/cost.txt
