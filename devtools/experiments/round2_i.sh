export TMPDIR=/tmp
O=$PWD/gpurun_out/r02i
mkdir -p $O
timeout 900 python -m pytest tests/test_presplit.py -m gpu -q 2>&1 | tail -30 > $O/pytest.txt
cat $O/pytest.txt
