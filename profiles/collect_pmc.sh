#!/bin/bash
# Counter passes for the sweep kernel (one rocprofv3 --pmc run per pass, no trace domains).
# usage: profiles/collect_pmc.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>_<pass>/
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 1 --warmup 0 --no-cpu --no-host-api --secondary off $*"
run() { # name counters...
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py $ARGS > gpurun_out/pmc_${tag}_${name}.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/pmc_${tag}_${name}.log; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
run sq2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT &&
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE TCC_EA0_ATOMIC_sum &&
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
