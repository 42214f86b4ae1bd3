#!/bin/bash
# What the library picks at each radius, and what it reaches (256^3, uniform; 10^4 sources up to R = 60, 1000 beyond)
for R in 3 5 8 10 12 15 20 25 30 35 40 45 50 60 70 80 90 100 120 128 140 160 200; do
  ns=10000; [ $R -gt 60 ] && ns=1000
  timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-host-api --secondary off --sources $ns --R $R 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['roofline']['launch']; print('| $R | $ns | %d | %s | %d | %d | %.3g | %.3g | %.0f %% |'%(l['block'], l['staging'], l.get('sectors_per_workgroup',1)*l.get('sources_per_workgroup',1), l['lds_bytes'], d['roofline']['kernel_ms'], d['value'], 100*d['roofline']['frac']))" || exit 1
done
