#!/bin/bash
# A/B of library builds in one GPU session: profiles/ab.sh <sources> lib1 lib2 ...
ns=$1; shift
for l in "$@"; do
  ASORA_HIP_LIB=$PWD/pyc2ray_amd/lib/$l.so timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --sources $ns 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', 'kernel_ms=%.2f'%d['roofline']['kernel_ms'], 'frac=%.3f'%d['roofline']['frac'], 'chk=%.10e'%d['config']['phi_checksum'])" || exit 1
done
