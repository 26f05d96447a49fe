export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_training.py -m gpu -q 2>&1 | tail -40
