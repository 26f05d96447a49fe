# Round-6 evidence run (on the MI355X box via gpurun), DEFAULT configuration of the tree: bash devtools/round_end_r06.sh TAG [notest]
export TMPDIR=/tmp
T=${1:-r06z}
O=$PWD/gpurun_out/$T
mkdir -p $O
if [ "$2" != "notest" ]; then
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -5 > $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
fi
# the headline line (full: verify + roofline + traffic PMC passes + CPU baseline + rows + box calibration), then the same-box
# A/B of the round's kernel: the folded down-sampling conv off / on / off (LC_FOLD_DOWN=0 = the reference's order of operations)
timeout 900 python bench.py 2>&1 | tail -1 > $O/bench.json
LC_FOLD_DOWN=0 timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-rows 2>&1 | tail -1 > $O/bench_fold0.json
timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-rows 2>&1 | tail -1 > $O/bench_fold1.json
LC_FOLD_DOWN=0 timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-rows 2>&1 | tail -1 > $O/bench_fold0b.json
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline --no-rows > $O/prof.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profc -o k -- python $GRAFT_REPO_ROOT/devtools/cond_run.py 8 12 > $O/profc.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o k -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 20 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline --no-rows > $O/prof1.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
timeout 900 python devtools/bench_rows.py > $O/rows.json 2> $O/rows.err
timeout 300 python devtools/fold_down_time.py 8 > $O/fold_down_time.txt 2>&1
cat $O/pytest.txt $O/smoke.txt 2>/dev/null; head -c 1200 $O/bench.json; echo; du -sh $O
