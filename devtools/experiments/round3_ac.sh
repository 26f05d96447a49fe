timeout 600 python -m pytest tests/test_range_safety.py tests/test_abi.py -q -m gpu -k "single_product" 2>&1 | grep -E "passed|failed|Error|assert" | head -5
python - <<'PY'
import sys, json, torch
sys.path.insert(0, "devtools")
import bench_rows as R
dev = torch.device("cuda:0")
print(json.dumps(R.uncond_autocast(dev, 8, 20)))
print(json.dumps(R.uncond(dev, 8, (32, 1024), 20, "uncond32")))
PY
