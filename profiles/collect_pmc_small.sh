#!/bin/bash
# Scheduler / wait counters for one bench configuration (three rocprofv3 --pmc passes, no trace domains).
# usage: profiles/collect_pmc_small.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>_<pass>/
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 1 --warmup 0 --no-cpu --no-host-api $*"
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py $ARGS > gpurun_out/pmc_${tag}_${name}.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/pmc_${tag}_${name}.log; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE &&
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS &&
run sq4 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F64 SQ_WAVES_EQ_64
