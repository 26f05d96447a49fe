S="8:64:64:32:1024:3"
echo "== prod"; python devtools/conv_bench.py --gn --emit --res $S
for v in n8 n24 n64 n88 n32; do echo "== $v"; python devtools/conv_bench.py --lib devtools/variants/liblc_$v.so --gn --emit --res $S; done
echo "== prod no res"; python devtools/conv_bench.py --gn --emit $S
