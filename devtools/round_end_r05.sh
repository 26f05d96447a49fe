# Round-5 evidence run (on the MI355X box via gpurun), DEFAULT configuration of the tree: bash devtools/round_end_r05.sh TAG [notest]
export TMPDIR=/tmp
T=${1:-r05zz2}
O=$PWD/gpurun_out/$T
mkdir -p $O
if [ "$2" != "notest" ]; then
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -3 > $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
fi
# the headline line (full: verify + roofline + traffic PMC passes + CPU baseline), then the same-box A/B of the round's kernel
timeout 900 python bench.py 2>&1 | tail -1 > $O/bench.json
LC_TALL=0 timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/bench_tall0.json
timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/bench_tall1.json
# ... and of the producer-side statistics (0 = a statistics pass in front of every GroupNorm), C2 and C3
LC_GN_PRODUCER_STATS=0 timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/bench_pstats0.json
LC_GN_PRODUCER_STATS=0 timeout 300 python devtools/bench_rows.py --only cond_layout_v6_32x1024 > $O/rows_c3_pstats0.json 2>/dev/null
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $O/prof.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profc -o k -- python $GRAFT_REPO_ROOT/devtools/cond_run.py 8 12 > $O/profc.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o k -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 20 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $O/prof1.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
timeout 900 python devtools/bench_rows.py > $O/rows.json 2> $O/rows.err
cat $O/pytest.txt $O/smoke.txt 2>/dev/null; head -c 1500 $O/bench.json; echo; du -sh $O
# PMC block: the level-0 launch (fused GroupNorm + residual + statistics, as in the C2 step; the heuristic's kernel = cfg 27)
rm -rf gpurun_out/pmcc; timeout 600 bash devtools/pmc_conv.sh 8 64 64 32 1024 3 0 --gn --emit --res > $O/pmc_level0.txt 2>&1
rm -rf gpurun_out/pmcc
tail -24 $O/pmc_level0.txt
