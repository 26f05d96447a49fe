export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
for lib in "" "--lib devtools/variants/liblc_nopub.so" "--lib devtools/variants/liblc_const.so"; do
  echo "== $lib" >> $O/conv_bench.txt
  timeout 300 python devtools/conv_bench.py $lib uncond8 >> $O/conv_bench.txt 2>&1
done
echo "== B1" >> $O/conv_bench.txt
timeout 300 python devtools/conv_bench.py uncond1 >> $O/conv_bench.txt 2>&1
cat $O/conv_bench.txt
