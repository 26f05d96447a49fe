export TMPDIR=/tmp
for mc in 64 0; do
  LC_FUSE_GN_MAX_CO=$mc timeout 300 python bench.py --no-cpu-baseline --no-verify --repeat 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fuse_max_co=$mc', d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'])"
done
