#!/bin/bash
# SQ (shader sequencer) counters of the bench's kernels, one small group per rocprofv3 pass (own runs, no
# tracing domains).  Runs on the GPU box via gpurun; tools/pmc_summary.py prints per-kernel averages.
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extras"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pass$i -o run -- $BENCH > $OUT/pass$i.json 2> $OUT/pass$i.log
  find $OUT/pass$i -name '*kernel_trace.csv' -delete
done
python $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -il "error\|invalid" $OUT/pass*.log | head
