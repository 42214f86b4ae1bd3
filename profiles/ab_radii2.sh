#!/bin/bash
# as ab_radii.sh with the source count per radius of radius_table.sh: profiles/ab_radii2.sh "<R list>" <lib1.so> <lib2.so> ...
radii=$1; shift
for R in $radii; do
  ns=10000; [ $R -gt 60 ] && ns=1000
  for lib in "$@"; do
    name=$(basename $lib .so)
    ASORA_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-host-api --R $R --sources $ns > gpurun_out/abr_${name}_$R.json 2> gpurun_out/abr_${name}_$R.err || { echo "$name R=$R failed"; tail -3 gpurun_out/abr_${name}_$R.err; exit 1; }
    python - <<PY
import json
j = json.loads(open("gpurun_out/abr_${name}_$R.json").read().strip().splitlines()[-1])
l = j["roofline"]["launch"]
print("R=$R", "$name", "kernel_ms=%.3f" % j["roofline"]["kernel_ms"], "frac=%.4f" % j["roofline"]["frac"], l["block"], l["staging"], l.get("sectors_per_workgroup"), l.get("sources_per_workgroup"))
PY
  done
done
