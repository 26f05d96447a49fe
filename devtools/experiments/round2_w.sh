for co in 64 128 256 512; do
  echo "== LC_FUSE_GN_MAX_CO=$co"
  LC_FUSE_GN_MAX_CO=$co python bench.py --no-cpu-baseline --no-verify --repeat 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B8', d['value'], d['ms_per_step'], d['roofline']['time_share_per_family_ms_per_step'])"
  LC_FUSE_GN_MAX_CO=$co python devtools/bench_rows.py --only uncond_32x1024,cond_layout_v6_32x1024 --quick 2>&1 | grep "\"batch\"\|ms_per_step"
done
