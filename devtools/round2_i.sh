export TMPDIR=/tmp
O=$PWD/gpurun_out/r02i
mkdir -p $O
timeout 900 python -m pytest tests/test_voxel_scatter.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
cat $O/pytest.txt
