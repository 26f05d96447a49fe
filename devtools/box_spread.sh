# one box: the headline (short form) + the C3 row + the box calibration.  bash devtools/box_spread.sh TAG
export TMPDIR=/tmp
T=${1:-spread}
O=$PWD/gpurun_out/$T
mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-rows 2>/dev/null | tail -1 > $O/bench.json
timeout 600 python devtools/bench_rows.py --quick --only cond_layout_v6_32x1024,uncond_32x1024 > $O/rows.json 2>/dev/null
python - <<PY
import json
d=json.load(open("$O/bench.json")); r=json.load(open("$O/rows.json"))
c=d.get("box_calibration") or {}
print("$T", "C2", d["value"], "steps/s", d["ms_per_step"], "ms | frac", d["roofline"]["frac"], "of box ceiling", d["roofline"].get("frac_of_box_ceiling"),
      "| bare MFMA", c.get("mfma_f16_tflops_random_operands"), "copy", c.get("stream_copy_tb_s"),
      "| C3 b8", [x["ms_per_step"] for x in r["cond_layout_v6_32x1024"]], "| uncond b1/b8", [x["ms_per_step"] for x in r["uncond_32x1024"]])
PY
