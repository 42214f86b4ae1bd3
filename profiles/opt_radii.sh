#!/bin/bash
# Launch-shape alternatives at the middle radii (256^3, 1e4 sources): kernel ms per option set.
# usage: profiles/opt_radii.sh "R R ..." "opts1" "opts2" ...    (opts: name=value,name=value or "default")
Rs=$1; shift
for R in $Rs; do
  for o in "$@"; do
    [ "$o" = default ] && export ASORA_OPTIONS= || export ASORA_OPTIONS=$o
    timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-host-api --secondary off --sources ${NS:-10000} --R $R 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['roofline']['launch']; print('R=$R %-44s %4d thr %-11s x%d  %6d B  %8.3f ms  %.0f %%'%('$o', l['block'], l['staging'], l.get('sectors_per_workgroup',1)*l.get('sources_per_workgroup',1), l['lds_bytes'], d['roofline']['kernel_ms'], 100*d['roofline']['frac']))" || echo "R=$R $o failed"
  done
done
