# round 5: vector x2 up-sampler -- full GPU suite on the final tree, then per-kernel times of the C2 and C3 steps and the headline
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05z22
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -4 | tee $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $O/prof.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profc -o k -- python $GRAFT_REPO_ROOT/devtools/cond_run.py 8 12 > $O/profc.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > $O/bench.json
timeout 300 python devtools/bench_rows.py --only cond_layout_v6_32x1024 2>/dev/null > $O/rows_c3.json
python - <<'PY'
import csv, json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05z22/"
for f in ("prof", "profc"):
    for r in csv.DictReader(open(O + f + "/k_kernel_stats.csv")):
        if "up2" in r["Name"] or "down2" in r["Name"]:
            print(f, r["Name"][:40], r["Calls"], round(float(r["TotalDurationNs"]) / 1e3 / int(r["Calls"]), 1), "us avg")
d = json.load(open(O + "bench.json")); print("c2", d["value"], d["ms_per_step"], d["verify"]["max_rel_l2_per_sample"])
print("c3", [r["ms_per_step"] for r in json.load(open(O + "rows_c3.json"))["cond_layout_v6_32x1024"]])
PY
