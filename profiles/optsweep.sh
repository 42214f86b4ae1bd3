# bench with a list of option settings: bash profiles/optsweep.sh "src_block=1 src_block=64" [bench args]
OPTS=$1; shift
for opt in $OPTS; do
ASORA_OPTIONS=$opt timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['roofline']['launch']; print('$opt', 'kernel_ms=%.3f'%d['roofline']['kernel_ms'], 'step_ms=%.3f'%d['ms_per_step'], 'frac=%.3f'%d['roofline']['frac'], 'value=%.3e'%d['value'], l['block'], l['staging'], l['lds_bytes'], d['config']['phi_checksum'])" || exit 1
done
