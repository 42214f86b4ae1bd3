#!/bin/bash
# Round 3, one GPU session: GPU tests, the bench line (with its `secondary` list), rocprofv3 kernel stats of the headline
# workload, the counter passes, the radius table and the configurations of SURVEY.md 8(d).
# usage (repo root on the GPU box): bash profiles/r3_session.sh <tag>;  then, in the container: python profiles/collect_evidence.py <tag>
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/${tag}_pytest.log 2>&1 || { tail -30 gpurun_out/${tag}_pytest.log; exit 1; }
tail -1 gpurun_out/${tag}_pytest.log
timeout -k 10 400 python -u bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
echo "bench ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --no-cpu --no-host-api --secondary off > gpurun_out/${tag}_prof.log 2>&1 || { tail -5 gpurun_out/${tag}_prof.log; exit 1; }
cat gpurun_out/${tag}_prof/*/*kernel_stats.csv | cut -c1-160 | head -5
bash profiles/collect_pmc.sh ${tag} --sources 10000 || exit 1
bash profiles/radius_table.sh > gpurun_out/${tag}_radius_rows.md || exit 1
echo "radius table ok"
bash profiles/config_table.sh > gpurun_out/${tag}_config_table.jsonl || exit 1
echo "config table ok"
