# round 4, call u: where the first sampling step of a process goes, with and without ops.prepare_model
mkdir -p gpurun_out/r04u
for args in "" "prepare" "cond" "cond prepare"; do
  timeout 200 python devtools/first_call.py $args 2>&1 | tail -1
done | tee gpurun_out/r04u/first_call.txt
