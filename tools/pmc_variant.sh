#!/bin/bash
# PMC counters for one kernel-library variant (tools/variants.py build ...) on a bench.py workload.
# usage: tools/pmc_variant.sh <variant name> <outdir-under-gpurun_out> [bench args...]
set -u
export BSLAM_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libbadslam_hip_$1.so; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_IFETCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-baseline 0 --pcg 0 --secondary 0 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
