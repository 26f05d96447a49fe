mkdir -p gpurun_out/r04i
timeout 120 python devtools/pp_phases.py 8:64:64:32:1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04i/phases.txt
timeout 120 python devtools/pp_phases.py 8:64:64:32:1024 --gn --emit --res 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04i/phases.txt
