# r03e: s_memtime phase totals of the level-0 conv (instrumented variant libraries; store cache policies)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03g; mkdir -p $O
{
for m in "" 0 3 4; do echo "== epi mode ${m:-2 (sc1)}"; LC_TIMING_LIB=liblc_timing$m.so python devtools/conv_phases.py 8:64:64:32:1024 --gn --res --emit; done
} > $O/out.txt 2>&1
cat $O/out.txt
