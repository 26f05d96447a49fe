for m in 0 1 0 1; do
LC_GN_PAIR_STATS=$m python - <<'PY'
import sys, os, json, torch
sys.path.insert(0, "devtools")
import bench_rows as R
dev = torch.device("cuda:0")
r = R.cond(dev, 8, 30); print("pair stats", os.environ["LC_GN_PAIR_STATS"], {k: r[k] for k in ("batch", "ms_per_step")})
PY
done
