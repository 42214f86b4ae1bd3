#!/bin/bash
# The synthetic configurations of SURVEY.md 8(d) through bench.py (one JSON line each): bash profiles/config_table.sh > out.jsonl
run() { timeout -k 10 300 python -u bench.py --no-cpu --no-host-api --secondary off --steps 3 --warmup 1 "$@" 2>/dev/null | tail -1 || exit 1; }
for ns in 1000 10000; do for R in 10 30 60; do run --sources $ns --R $R; done; done
run --xh 1e-4
run --dr 6.0268e17            # 50 pc / 256: optically thin cells
run --density clustered --xh 1e-4
run --N 128 --sources 1 --R 150 --dr 3.375e20 --numtau 10000 --steps 20   # configs[0]-like: single source, full box
run --N 250 --sources 12500 --R 58.6 --density clustered                 # eor_250 per-GPU share of 1e5 sources
run --N 512 --sources 2000 --R 60 --density clustered                    # stress_512 (capacity; per-GPU share is 125 000)
run --N 512 --sources 500 --R 150 --density clustered
