#!/bin/bash
# A/B of library builds on one box: profiles/ab_bench.sh <out-prefix> <lib1.so> <lib2.so> ...  [env: AB_ARGS="--R 60 ..."]
# Each build runs bench.py twice, interleaved (boxes drift by ~1-2 %); kernel_ms of each run is printed.
pfx=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    ASORA_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-host-api $AB_ARGS > gpurun_out/${pfx}_${name}_$rep.json 2> gpurun_out/${pfx}_${name}_$rep.err || exit 1
    python - <<PY
import json
j = json.loads(open("gpurun_out/${pfx}_${name}_$rep.json").read().strip().splitlines()[-1])
print("$name", "rep$rep", "kernel_ms=%.3f" % j["roofline"]["kernel_ms"], "frac=%.4f" % j["roofline"]["frac"], "value=%.4g" % j["value"], j["roofline"]["launch"]["block"], j["roofline"]["launch"]["staging"])
PY
  done
done
