# First GPU call of the next round: does the four-32-bit-store form of the statistics entry cure the corruption?
# (profiles/r03_conv_phases.txt, last two sections.)  Builds a variant library, then runs the per-entry stress test and
# the statistics parity tests against it five times with the producer statistics ON.
# (build the variants in the build container BEFORE the gpurun call -- the .so files travel; mode 1 = four 32-bit stores,
#  mode 2 = one 128-bit store without SGPR soffset, for which hipcc inserts the ISA's wait state itself)
#   cd lidarcrafter_amd; objs=$(ls build/*.o | grep -v "conv_f16x2.o\|conv_f16x2_p1.o"); mkdir -p ../devtools/variants
#   for m in 1 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLC_ENTRY_DWORD_STORES=$m -c csrc/conv_f16x2.hip -o /tmp/c$m.o
#     hipcc --offload-arch=gfx950 -shared -fPIC -o ../devtools/variants/liblc_entry_m$m.so /tmp/c$m.o $objs; done
M=${1:-1}
for i in 1 2 3 4 5; do
  LC_HIP_LIB=devtools/variants/liblc_entry_m$M.so LC_GN_PRODUCER_STATS=1 timeout 300 python -m pytest tests -m gpu -q -rf \
      -k "under_load or stats or producer or split_k or concat_segments or c2_ddim50 or composed" 2>&1 | grep -E "^FAILED|passed|failed" | tr "\n" " "; echo
done
# same-box cost of the change on the emitting level-0 launch and on the step
for lib in lidarcrafter_amd/liblidarcrafter_hip.so devtools/variants/liblc_entry_m$M.so; do
  LC_HIP_LIB=$lib python devtools/conv_time.py 8:64:64:32:1024 --gn --emit --res 2>&1 | grep us
  LC_HIP_LIB=$lib LC_GN_PRODUCER_STATS=1 python bench.py --steps 20 --warmup 5 --repeat 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c100-260
done
