export TMPDIR=/tmp
O=$PWD/gpurun_out/r02g
mkdir -p $O
S="8:512:256:4:128:1 8:512:128:8:256:1 8:256:64:16:512:1 8:128:64:32:1024:1 8:512:1536:4:128:1 8:512:512:4:128:1 8:256:256:8:256:3 8:64:64:32:1024:3"
echo "== old" >> $O/mb.txt
LC_TREE=$PWD/devtools/old_r1 timeout 300 python devtools/conv_bench.py $S >> $O/mb.txt 2>&1
echo "== new" >> $O/mb.txt
timeout 300 python devtools/conv_bench.py $S >> $O/mb.txt 2>&1
echo "== old" >> $O/mb.txt
LC_TREE=$PWD/devtools/old_r1 timeout 300 python devtools/conv_bench.py $S >> $O/mb.txt 2>&1
cat $O/mb.txt
