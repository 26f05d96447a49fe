export TMPDIR=/tmp
O=$PWD/gpurun_out/r06y; mkdir -p $O
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 10 --warmup 3 --repeat 1 --no-verify --no-cpu-baseline --no-roofline --no-rows > $O/prof.log 2>&1)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python devtools/trace_seq.py $f pstep > $O/seq_b1.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
grep -o '"ms_per_step": [0-9.]*' $O/prof.log | head -1
