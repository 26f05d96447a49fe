# Round-2 evidence run (on the MI355X box via gpurun): bash devtools/round_end_r02.sh TAG
export TMPDIR=/tmp
T=${1:-r02a}
O=$PWD/gpurun_out/$T
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
timeout 500 python bench.py 2>&1 | tail -1 > $O/bench.json
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $O/prof.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $O/pf.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $O/pw.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
timeout 500 python devtools/bench_rows.py > $O/rows.json 2> $O/rows.err
cat $O/pytest.txt $O/smoke.txt; head -c 400 $O/bench.json; echo; du -sh $O
