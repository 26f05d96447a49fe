# experiment: store forms of the 1x1 pre-split kernel's plain epilogue (LC_P1_ST 0..3) x block forms (LC_P1_BN)
export TMPDIR=/tmp
T=${1:-r06u5}
O=$PWD/gpurun_out/$T
mkdir -p $O
for st in 2 3; do
LC_P1_ST=$st timeout 600 python -m pytest tests/test_fold_up.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 > $O/pytest_st$st.txt
done
for st in 0 1 2 3; do
LC_P1_ST=$st LC_FOLD_UP_MIN_CI=32 timeout 600 python devtools/fold_up_time.py 8 > $O/time_wide_st$st.txt 2>&1
LC_P1_ST=$st LC_P1_BN=128 LC_FOLD_UP_MIN_CI=32 timeout 600 python devtools/fold_up_time.py 8 > $O/time_narrow_st$st.txt 2>&1
done
cd $O; tail -2 pytest_st*.txt; grep -H "planes" time_*.txt | sed -e 's/reference order.*folded/folded/' -e 's/| rel-L2.*//'
