# round 5: clean A/B of the write-through (sc1) stores in the resampling kernels, the attention output, the fp32 GroupNorm apply
# kernels and the non-pipelined conv kernel: shipped library against devtools/variants/liblc_wt0.so (all of them write-back);
# both with the step's verification on
# (the variant: resample.hip / attention.hip / norm.hip compiled with -DLC_ST_WT=0 and conv_f16x2.hip with -DLC_NP_AUX=0, linked with the
#  other objects of lidarcrafter_amd/build/ -- devtools/build_var.sh builds one-TU variants the same way; earlier experiments of the
#  z series reused this file: their outputs are in profiles/r05_second_half_raw.txt)
export TMPDIR=/tmp
O=gpurun_out/r05z20
mkdir -p $O
for v in wt0 new wt0 new; do
  if [ $v = new ]; then unset LC_HIP_LIB; else export LC_HIP_LIB=devtools/variants/liblc_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-traffic --repeat 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 $v', d['value'], d['ms_per_step'], d.get('verify', d.get('verification')))" | tee -a $O/step.txt
  timeout 300 python devtools/bench_rows.py --only cond_layout_v6_32x1024 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['cond_layout_v6_32x1024']; print('c3 $v', [r['ms_per_step'] for r in d])" | tee -a $O/step.txt
done
