# R sweep at 1000 sources: bash profiles/rsweep.sh "60 80 100" "wg_threads=256 wg_threads=1024"
RS=${1:-"80 100 140"}; OPTS=${2:-"wg_threads=0"}
for R in $RS; do for opt in $OPTS; do
ASORA_OPTIONS=$opt timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu --sources 1000 --R $R 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R=$R $opt', 'kernel_ms=%.2f'%d['roofline']['kernel_ms'], 'frac=%.3f'%d['roofline']['frac'], d['roofline']['launch'], 'value=%.3e'%d['value'])" || exit 1
done; done
