S="8:256:768:8:256:1 8:512:1536:4:128:1 8:256:256:8:256:1 8:512:512:4:128:1 8:128:384:16:512:1"
for c in 0 1 2 4; do echo "== cfg $c"; python devtools/conv_bench.py --cfg $c $S; done
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "roi or conv_golden" 2>&1 | grep -E "passed|failed"
