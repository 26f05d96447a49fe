export TMPDIR=/tmp
mkdir -p gpurun_out/r01i
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r01i/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r01i/smoke.txt
timeout 400 python bench.py 2>&1 | tail -1 > gpurun_out/r01i/bench.json
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01i/prof -o r01i -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r01i/prof.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01i/pf -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r01i/pf.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01i/pw -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r01i/pw.log 2>&1)
find gpurun_out/r01i -name "*kernel_trace.csv" -delete
timeout 400 python devtools/bench_rows.py > gpurun_out/r01i/rows.json 2> gpurun_out/r01i/rows.err
cat gpurun_out/r01i/pytest.txt gpurun_out/r01i/smoke.txt gpurun_out/r01i/bench.json; du -sh gpurun_out/r01i
