# round 5: SiLU with the hardware reciprocal + folded affine in the pre-split GroupNorm apply pass; the pre-split conv's
# epilogue as buffer operations (A/B against the predicated form: devtools/variants/liblc_bufepi0.so)
export TMPDIR=/tmp
O=gpurun_out/r05z7
mkdir -p $O
timeout 1500 python -m pytest tests/test_presplit.py tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "presplit or resample or groupnorm or gn or unet or conv_pp or statistics or pair or linear or silu" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -5 | tee $O/pytest_a.txt
{ for v in new old new old; do
    if [ $v = old ]; then export LC_HIP_LIB=devtools/variants/liblc_bufepi0.so; else unset LC_HIP_LIB; fi
    echo "-- $v"; timeout 120 python devtools/ps_time.py 8 --emit 8; done; unset LC_HIP_LIB; } 2>&1 | grep "^ps\|^--" | tee $O/ps_bufepi.txt
for i in 1 2; do
  LC_HIP_LIB=devtools/variants/liblc_bufepi0.so timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/c2_old_$i.json
  timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>&1 | tail -1 > $O/c2_new_$i.json
done
timeout 300 python devtools/bench_rows.py --only cond_layout_v6_32x1024 2>/dev/null > $O/c3_new.json
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05z7/c3_*.json")):
    d=json.load(open(f))["cond_layout_v6_32x1024"]; print(f.split('/')[-1], [r["ms_per_step"] for r in d])
for f in sorted(glob.glob("gpurun_out/r05z7/c2_*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], d["value"], d["ms_per_step"])
PY
