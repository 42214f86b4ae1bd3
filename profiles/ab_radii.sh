#!/bin/bash
# A/B of library builds over radii: profiles/ab_radii.sh "<R list>" <sources> <lib1.so> <lib2.so> ...   (N = 256, uniform grid)
radii=$1; ns=$2; shift 2
for R in $radii; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    ASORA_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-host-api --R $R --sources $ns > gpurun_out/abr_${name}_$R.json 2> gpurun_out/abr_${name}_$R.err || { echo "$name R=$R failed"; tail -3 gpurun_out/abr_${name}_$R.err; exit 1; }
    python - <<PY
import json
j = json.loads(open("gpurun_out/abr_${name}_$R.json").read().strip().splitlines()[-1])
print("R=$R", "$name", "kernel_ms=%.3f" % j["roofline"]["kernel_ms"], "frac=%.4f" % j["roofline"]["frac"], j["roofline"]["launch"]["block"], j["roofline"]["launch"]["staging"])
PY
  done
done
