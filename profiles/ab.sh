#!/bin/bash
# A/B of library builds / options in one GPU session:
#   profiles/ab.sh <sources> lib1[:opt=val,...] lib2 ...     (libs under pyc2ray_amd/lib/, without .so)
#   ASORA_AB_ARGS="--heat" adds bench arguments
ns=$1; shift
for spec in "$@"; do
  l=${spec%%:*}; o=""; [[ "$spec" == *:* ]] && o=${spec#*:}
  ASORA_OPTIONS=$o ASORA_HIP_LIB=$PWD/pyc2ray_amd/lib/$l.so timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --sources $ns $ASORA_AB_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', 'kernel_ms=%.2f'%d['roofline']['kernel_ms'], 'frac=%.3f'%d['roofline']['frac'], 'lds=%d'%d['roofline']['launch']['lds_bytes'], 'chk=%.10e'%d['config']['phi_checksum'])" || exit 1
done
