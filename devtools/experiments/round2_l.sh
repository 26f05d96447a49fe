bash devtools/pmc_ps.sh ps256 --ps 8:256:256:8:256:3 > gpurun_out/pmc_ps256.txt 2>&1
bash devtools/pmc_ps.sh fp256 8:256:256:8:256:3 > gpurun_out/pmc_fp256.txt 2>&1
cat gpurun_out/pmc_ps256.txt; echo; cat gpurun_out/pmc_fp256.txt
