#!/bin/bash
# A/B of library builds over radii (256^3, 1e4 sources up to R = 60): kernel ms per build.
# usage: profiles/ab_radii3.sh "R R ..." base tag1 tag2 ...
Rs=$1; shift
for R in $Rs; do
  line="R=$R"
  for t in "$@"; do
    if [ $t = base ]; then unset ASORA_HIP_LIB; else export ASORA_HIP_LIB=$PWD/ab/lib_$t.so; fi
    ns=10000; [ ${R%.*} -gt 60 ] && ns=1000
    ms=$(timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-host-api --secondary off --sources $ns --R $R 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['roofline']['kernel_ms'])")
    line="$line  $t=$ms"
  done
  echo "$line"
done
