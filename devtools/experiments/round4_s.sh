# round 4, call s: N consecutive full GPU suites of the final tree (pass / fail lines kept), each under its own timeout
mkdir -p gpurun_out/r04s
N=${1:-6}
for i in $(seq 1 $N); do
  timeout 400 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tr "\n" " "; echo
done | tee gpurun_out/r04s/suite.txt
