export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.txt
timeout 500 python bench.py 2>&1 | tail -1 > $O/bench.json
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r02a -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
timeout 600 python devtools/passes_error.py > $O/passes_error.json 2> $O/passes_error.err
timeout 400 python devtools/bench_rows.py > $O/rows.json 2> $O/rows.err
cat $O/pytest.txt $O/smoke.txt $O/bench.json; du -sh $O
