export TMPDIR=/tmp
T=${1:-r06z}
O=$PWD/gpurun_out/$T
mkdir -p $O
timeout 900 python -m pytest tests/test_fold_up.py tests/test_bench_shapes.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 > $O/pytest.txt
for v in 0 1 0 1; do
LC_FOLD_UP_X=$v timeout 600 python devtools/bench_rows.py --quick --only cond_layout_v6_32x1024 > $O/rows_x$v.$RANDOM.json 2>> $O/rows.err
done
cat $O/pytest.txt; for f in $O/rows_x*.json; do echo -n "$f "; grep -h "ms_per_step" $f | tr -d '\n'; echo; done
