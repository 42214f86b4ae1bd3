#!/bin/bash
# A/B of library builds on one box: the sweep-kernel time of a bench workload for the in-tree library ("base") and for
# every ab/lib_<tag>.so named.   usage: profiles/ab_libs.sh "<bench args>" base tag1 tag2 ...
args=$1; shift
for t in "$@"; do
  if [ $t = base ]; then unset ASORA_HIP_LIB; else export ASORA_HIP_LIB=$PWD/ab/lib_$t.so; fi
  timeout -k 10 200 python bench.py --no-cpu --no-host-api --secondary off --steps 4 $args > gpurun_out/ab_$t.json 2>/dev/null || { echo "$t failed"; continue; }
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab_{t}.json").read().strip().splitlines()[-1])
print(t, "kernel_ms", round(d["roofline"]["kernel_ms"], 2), "frac", round(d["roofline"]["frac"], 3), "checksum", d["config"]["phi_checksum"])
PY
done
