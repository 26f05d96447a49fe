export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_shapes.py -m gpu -q -k "harness or second" 2>&1 | tail -30
