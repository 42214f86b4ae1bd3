for R in 80 100 140; do for lim in 0; do
ASORA_OPTIONS=lds_stage_limit=$lim timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu --sources 1000 --R $R 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R=$R lim=$lim', 'kernel_ms=%.2f'%d['roofline']['kernel_ms'], 'frac=%.3f'%d['roofline']['frac'], d['roofline']['launch'], 'value=%.3e'%d['value'])" || exit 1
done; done
