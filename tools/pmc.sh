#!/bin/bash
# Collects PMC counters for bench.py in separate passes (never combined with tracing).
# usage: tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-baseline 0 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
ls -R $OUT | head -30
