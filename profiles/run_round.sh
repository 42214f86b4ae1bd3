#!/bin/bash
# One GPU session: parity tests, bench line, rocprofv3 kernel stats, PMC passes.
# usage (from the repo root on the GPU box): bash profiles/run_round.sh <tag>
# Results land in gpurun_out/<tag>_*; copy the summaries to profiles/ afterwards.
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/${tag}_pytest.log 2>&1 || { tail -30 gpurun_out/${tag}_pytest.log; exit 1; }
tail -3 gpurun_out/${tag}_pytest.log
timeout -k 10 400 python -u bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
cat gpurun_out/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --no-cpu --no-host-api > gpurun_out/${tag}_prof.log 2>&1 || { tail -5 gpurun_out/${tag}_prof.log; exit 1; }
cat gpurun_out/${tag}_prof/*/*kernel_stats.csv | cut -c1-200 | head -6
bash profiles/collect_pmc.sh ${tag} --sources 10000
