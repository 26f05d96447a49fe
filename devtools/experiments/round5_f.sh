# round 5: PMC of the tall kernel (cfg 27) on the level-0 shape, full and plain, next to cfg 23
export TMPDIR=/tmp
O=gpurun_out/r05f
mkdir -p $O
rm -rf gpurun_out/pmcc; timeout 400 bash devtools/pmc_conv.sh 8 64 64 32 1024 3 27 --gn --emit --res > $O/pmc_tall_full.txt 2>&1
rm -rf gpurun_out/pmcc; timeout 400 bash devtools/pmc_conv.sh 8 64 64 32 1024 3 27 > $O/pmc_tall_plain.txt 2>&1
rm -rf gpurun_out/pmcc; timeout 400 bash devtools/pmc_conv.sh 8 64 64 32 1024 3 23 > $O/pmc_pipe_plain.txt 2>&1
rm -rf gpurun_out/pmcc
paste $O/pmc_tall_full.txt $O/pmc_tall_plain.txt $O/pmc_pipe_plain.txt | awk '{print $1, $3, $6, $9}' | column -t
