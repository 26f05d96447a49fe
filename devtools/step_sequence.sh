#!/bin/bash
# Kernel sequence of ONE replayed denoising step with durations (rocprofv3 --kernel-trace; the trace itself is deleted):
#   bash devtools/step_sequence.sh TAG            -> C2 (bench.py's step)         gpurun_out/TAG/seq.txt
#   bash devtools/step_sequence.sh TAG cond [B]   -> C3 shape (devtools/cond_run.py) gpurun_out/TAG/seq_cond.txt
export TMPDIR=/tmp
T=${1:-seq}; O=$PWD/gpurun_out/$T; mkdir -p $O
if [ "$2" == "cond" ]; then
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/profc -o k -- python $GRAFT_REPO_ROOT/devtools/cond_run.py ${3:-8} 8 > $O/profc.log 2>&1)
  f=$(find $O/profc -name "*kernel_trace.csv" | head -1); out=$O/seq_cond.txt
else
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline --no-rows > $O/prof.log 2>&1)
  f=$(find $O/prof -name "*kernel_trace.csv" | head -1); out=$O/seq.txt
fi
python devtools/trace_seq.py $f pstep > $out 2>&1
find $O -name "*kernel_trace.csv" -delete
tail -3 $out
