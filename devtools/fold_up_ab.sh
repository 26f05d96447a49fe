# Up-path fold (LC_FOLD_UP): parity tests, per-shape timing, same-box A/B of the C3 / C2 steps.  bash devtools/fold_up_ab.sh TAG [quick|notime]
export TMPDIR=/tmp
T=${1:-r06u}
O=$PWD/gpurun_out/$T
mkdir -p $O
if [ "$2" != "quick" ]; then
timeout 900 python -m pytest tests/test_fold_up.py tests/test_presplit.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_fold_up.txt
timeout 900 python -m pytest tests/test_bench_shapes.py tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "c3_b8 or c2_forward or c2_ddim50 or qkv or units or cond_" 2>&1 | tail -15 > $O/pytest_shapes.txt
fi
LC_FOLD_UP_MIN_CI=32 timeout 600 python devtools/fold_up_time.py 8 > $O/fold_up_time.txt 2>&1
if [ "$2" != "notime" ]; then
for i in 0 1 0 1; do
LC_FOLD_UP=$i timeout 600 python devtools/bench_rows.py --quick --only cond_layout_v6_32x1024,uncond_32x1024 > $O/rows_fold$i.$RANDOM.json 2>> $O/rows.err
done
LC_FOLD_UP_MIN_CI=256 timeout 600 python devtools/bench_rows.py --quick --only cond_layout_v6_32x1024,uncond_32x1024 > $O/rows_fold1_min256.json 2>> $O/rows.err
LC_P1_ST=0 timeout 600 python devtools/bench_rows.py --quick --only cond_layout_v6_32x1024,uncond_32x1024 > $O/rows_fold1_st0.json 2>> $O/rows.err
fi
cat $O/pytest_fold_up.txt $O/pytest_shapes.txt $O/fold_up_time.txt 2>/dev/null; for f in $O/rows_fold*.json; do echo $f; grep -h "ms_per_step" $f | tr -d '\n'; echo; done
