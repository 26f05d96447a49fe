export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_object_branch.py tests/test_presplit.py tests/test_range_safety.py -m gpu -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --repeat 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'])"
