# three products vs one product per multiply, same kernels otherwise: how MFMA-bound is each conv?
for p in 3 1; do echo "== products $p"
LC_CONV_PRODUCTS=$p python devtools/ps_time.py 8 2>&1 | grep "^ps"
LC_CONV_PRODUCTS=$p python devtools/conv_time.py 8:64:64:32:1024 8:128:64:32:1024 --gn --emit --res 2>&1 | grep us
done
