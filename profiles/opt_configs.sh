#!/bin/bash
# One library, several option sets per configuration:
#   profiles/opt_configs.sh <out-prefix> "<opts1> <opts2> ..." "<bench args 1>" ...   (an option set: name=value,name=value or "-")
pfx=$1; optsets=$2; shift 2
i=0
for cfg in "$@"; do
  i=$((i+1)); k=0
  for o in $optsets; do
    k=$((k+1)); oo=$o; [ "$o" = "-" ] && oo=""
    ASORA_OPTIONS=$oo timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-host-api $cfg > gpurun_out/${pfx}_c${i}_o${k}.json 2> gpurun_out/${pfx}_c${i}_o${k}.err || { echo "FAILED [$cfg] $o"; tail -2 gpurun_out/${pfx}_c${i}_o${k}.err; continue; }
    python - <<PY
import json
j = json.loads(open("gpurun_out/${pfx}_c${i}_o${k}.json").read().strip().splitlines()[-1])
l = j["roofline"]["launch"]
print("[$cfg]", "$o", "kernel_ms=%.3f" % j["roofline"]["kernel_ms"], "frac=%.4f" % j["roofline"]["frac"], l["block"], l["staging"], l["sectors_per_workgroup"], l.get("sources_per_workgroup"), l["lds_bytes"])
PY
  done
done
