#!/bin/bash
# Every rank's share of the bench at 1 / 2 / 4 / 8 GPUs, rehearsed on ONE GPU (ASORA_BENCH_EMULATE=rank/world: the
# curve-sorted list sharded as an N-GPU run shards it, this GPU sweeps one shard; no collective).
for w in 1 2 4 8; do
  for r in $(seq 0 $((w-1))); do
    ASORA_BENCH_EMULATE=$r/$w timeout -k 10 120 python bench.py --no-cpu --no-host-api --secondary off --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('world $w rank $r kernel_ms=%.3f step_ms=%.3f %d' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['config']['sources_per_gpu']))" || exit 1
  done
done
