#!/bin/bash
# A/B of library builds over several configurations on one box:
#   profiles/ab_configs.sh <out-prefix> "<lib1.so> <lib2.so> ..." "<bench args of config 1>" "<bench args of config 2>" ...
pfx=$1; libs=$2; shift 2
i=0
for cfg in "$@"; do
  i=$((i+1))
  for lib in $libs; do
    name=$(basename $lib .so)
    ASORA_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-host-api $cfg > gpurun_out/${pfx}_c${i}_${name}.json 2> gpurun_out/${pfx}_c${i}_${name}.err || { echo "FAILED $name $cfg"; tail -3 gpurun_out/${pfx}_c${i}_${name}.err; exit 1; }
    python - <<PY
import json
j = json.loads(open("gpurun_out/${pfx}_c${i}_${name}.json").read().strip().splitlines()[-1])
l = j["roofline"]["launch"]
print("[$cfg]", "$name", "kernel_ms=%.3f" % j["roofline"]["kernel_ms"], "frac=%.4f" % j["roofline"]["frac"], "value=%.4g" % j["value"], l["block"], l["staging"], l["sectors_per_workgroup"], l["lds_bytes"])
PY
  done
done
