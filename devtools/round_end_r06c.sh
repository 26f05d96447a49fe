# Round-6 evidence run, third part (up-path fold + 16-byte store form of the 1x1 projections), DEFAULT configuration of the tree:
#   bash devtools/round_end_r06c.sh TAG [notest]
export TMPDIR=/tmp
T=${1:-r06c}
O=$PWD/gpurun_out/$T
mkdir -p $O
if [ "$2" != "notest" ]; then
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -5 > $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
fi
# the headline line (full: verify + roofline + traffic PMC passes + CPU baseline + rows + box calibration), then same-box A/Bs:
# the round's third part off (LC_FOLD_UP=0 LC_PS1X1_MIN_CO=100000 LC_P1_ST=0 = the second half's tree) / on / off
timeout 900 python bench.py 2>&1 | tail -1 > $O/bench.json
B="timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-rows"
LC_FOLD_UP=0 LC_PS1X1_MIN_CO=100000 LC_P1_ST=0 $B 2>&1 | tail -1 > $O/bench_part3_off.json
$B 2>&1 | tail -1 > $O/bench_part3_on.json
LC_FOLD_UP=0 LC_PS1X1_MIN_CO=100000 LC_P1_ST=0 $B 2>&1 | tail -1 > $O/bench_part3_off_b.json
LC_FOLD_DOWN=0 LC_FOLD_UP=0 $B 2>&1 | tail -1 > $O/bench_folds_off.json
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline --no-rows > $O/prof.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profc -o k -- python $GRAFT_REPO_ROOT/devtools/cond_run.py 8 12 > $O/profc.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
timeout 900 python devtools/bench_rows.py > $O/rows.json 2> $O/rows.err
for i in 0 1 0 1; do
LC_FOLD_UP=$i timeout 600 python devtools/bench_rows.py --quick --only cond_layout_v6_32x1024 > $O/rows_c3_fold$i.$RANDOM.json 2>> $O/rows.err
done
LC_FOLD_UP=0 LC_P1_ST=0 timeout 600 python devtools/bench_rows.py --quick --only cond_layout_v6_32x1024 > $O/rows_c3_part3_off.json 2>> $O/rows.err
bash devtools/step_sequence.sh $T > /dev/null 2>&1
bash devtools/step_sequence.sh $T cond > /dev/null 2>&1
cat $O/pytest.txt $O/smoke.txt 2>/dev/null; head -c 1200 $O/bench.json; echo; du -sh $O
