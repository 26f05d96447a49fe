# round 5: PMC block of the pre-split 3x3 kernel (the step's second kernel family) on two C2 / C3 layer shapes
export TMPDIR=/tmp
O=gpurun_out/r05z21
mkdir -p $O
for sh in 0 3; do
  rm -rf gpurun_out/pmcc
  PMC_SCRIPT=devtools/ps_time.py PMC_KERNEL=conv_f16x2_ps_kernel timeout 500 bash devtools/pmc_conv.sh 8 --emit 8 --shape $sh > $O/pmc_ps_shape$sh.txt 2>&1
  grep "^ps" gpurun_out/pmcc/log1.txt >> $O/pmc_ps_shape$sh.txt
done
rm -rf gpurun_out/pmcc
tail -30 $O/pmc_ps_shape0.txt
