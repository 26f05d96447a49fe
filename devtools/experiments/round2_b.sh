export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout 1500 python -m pytest tests/test_range_safety.py tests/test_bench_shapes.py -m gpu -q 2>&1 | tail -40 > $O/pytest.txt
timeout 500 python bench.py 2>&1 | tail -1 > $O/bench.json
timeout 400 python devtools/bench_rows.py > $O/rows.json 2> $O/rows.err
tail -5 $O/pytest.txt; cat $O/bench.json; du -sh $O
