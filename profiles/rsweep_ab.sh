# radius sweep for two library builds: bash profiles/rsweep_ab.sh "30 60 90" libA libB   (libs under pyc2ray_amd/lib/, without .so)
RS=$1; shift
for R in $RS; do for l in "$@"; do
ASORA_HIP_LIB=$PWD/pyc2ray_amd/lib/$l.so timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --sources 2000 --R $R 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R=$R $l', 'kernel_ms=%.2f'%d['roofline']['kernel_ms'], 'frac=%.3f'%d['roofline']['frac'], d['roofline']['launch']['block'], d['roofline']['launch']['staging'])" || exit 1
done; done
