#!/bin/bash
# the 512^3 / R = 150 configuration (500 clustered sources) for each library given: profiles/ab_n512.sh <lib1.so> ...
for l in "$@"; do
  ASORA_HIP_LIB=$PWD/$l timeout -k 10 300 python bench.py --no-cpu --no-host-api --steps 3 --warmup 1 --N 512 --sources 500 --R 150 --density clustered > gpurun_out/abn512.json 2> gpurun_out/abn512.err || { echo "$l failed"; tail -3 gpurun_out/abn512.err; exit 1; }
  python - <<PY
import json
j = json.loads(open("gpurun_out/abn512.json").read().strip().splitlines()[-1])
print("N512R150", "$l", "kernel_ms=%.3f" % j["roofline"]["kernel_ms"], "frac=%.4f" % j["roofline"]["frac"], j["roofline"]["launch"]["block"])
PY
done
