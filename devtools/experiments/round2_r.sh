# level-0 ablation: 64->64 and 128->64 @32x1024 B=8, fused GN + emit (+res)
S="8:64:64:32:1024:3 8:128:64:32:1024:3"
echo "== prod plain";     python devtools/conv_bench.py $S
echo "== prod gn";        python devtools/conv_bench.py --gn $S
echo "== prod gn emit";   python devtools/conv_bench.py --gn --emit $S
echo "== prod gn emit res"; python devtools/conv_bench.py --gn --emit --res $S
for v in abl1 abl2 abl3 abl8 abl16 abl24 abl32; do
  echo "== $v gn emit res"; python devtools/conv_bench.py --lib devtools/variants/liblc_$v.so --gn --emit --res $S
done
echo "== ps emit res";  python devtools/conv_bench.py --ps --emit --res $S
