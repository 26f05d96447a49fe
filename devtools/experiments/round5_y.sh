# round 5: the C3 step's GroupNorm statistics lookups, with what each statistics-pass tensor does carry
export TMPDIR=/tmp
O=gpurun_out/r05y
mkdir -p $O
for s in 1 2; do LC_GN_TRACE=1 timeout 300 python devtools/cond_run.py 8 $s 2>&1 | grep -E "gn lookup|ok" > $O/gn_trace_c3_s$s.txt; done
