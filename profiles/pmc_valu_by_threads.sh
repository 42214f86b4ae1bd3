#!/bin/bash
# VALU instructions of the sweep at R = 60 for several workgroup sizes (same cells, different numbers of phases):
# the difference per (wave, phase) pair is the fixed cost of a phase.  One rocprofv3 --pmc pass per size.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in 128 192 256 512; do
  ASORA_OPTIONS=wg_threads=$t timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_thr_$t -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-host-api --sources 2000 > gpurun_out/pmc_thr_$t.log 2>&1 || { echo "pass $t failed"; tail -3 gpurun_out/pmc_thr_$t.log; exit 1; }
  echo "pass $t ok"
done
