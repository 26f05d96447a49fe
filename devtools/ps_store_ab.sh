# 16-byte store form of the pre-split 3x3 kernel's open epilogue (LC_PS_ST): parity, then same-box A/B of the C2 / C3 steps
export TMPDIR=/tmp
T=${1:-r06v}
O=$PWD/gpurun_out/$T
mkdir -p $O
timeout 1200 python -m pytest tests/test_presplit.py tests/test_fold_up.py tests/test_bench_shapes.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $O/pytest.txt
for i in 0 1 0 1; do
LC_PS_ST=$i timeout 600 python devtools/bench_rows.py --quick --only cond_layout_v6_32x1024,uncond_32x1024 > $O/rows_st$i.$RANDOM.json 2>> $O/rows.err
done
timeout 300 python devtools/ps_time.py > $O/ps_time_st1.txt 2>&1
LC_PS_ST=0 timeout 300 python devtools/ps_time.py > $O/ps_time_st0.txt 2>&1
cat $O/pytest.txt; for f in $O/rows_st*.json; do echo $f; grep -h "ms_per_step" $f | tr -d '\n'; echo; done; tail -12 $O/ps_time_st0.txt $O/ps_time_st1.txt
