# round 4, call h: PMC of the ping-pong kernel (cfg 33) next to cfg 23, level-0 shape with GN + statistics + residual
export TMPDIR=/tmp
mkdir -p gpurun_out/r04h
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|name)|TA_|TCP_|TCC_|SQ_INSTS_VMEM|SQ_WAIT_INST|SQ_ACTIVE_INST|SQC_" | head -400 > gpurun_out/r04h/avail.txt
for cfg in 33 23; do
  rm -rf gpurun_out/pmcc
  bash devtools/pmc_conv.sh 8:64:64:32:1024 --gn --emit --res --cfg $cfg > gpurun_out/r04h/pmc_cfg$cfg.txt 2>&1
done
tail -30 gpurun_out/r04h/pmc_cfg33.txt
