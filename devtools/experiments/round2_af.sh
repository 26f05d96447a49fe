S="8:128:128:16:512:3 8:256:256:8:256:3 8:512:512:4:128:3"
echo "== ps";       python devtools/conv_bench.py --ps $S
echo "== ps emit";  python devtools/conv_bench.py --ps --emit $S
for v in e1 e2 e3; do echo "== ps emit $v"; python devtools/conv_bench.py --lib devtools/variants/liblc_$v.so --ps --emit $S; done
