one() { env "$@" python bench.py --no-legs $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$* $EXTRA', round(d['value']/1e6,1), 'M/s  exp in-region', round(d.get('roofline',{}).get('avg_launch_ms',0),3), 'subchain', round(d['kernels'][0]['avg_launch_ms'],3))"; }
for rep in 1 2; do one BSX_FUSED_HINT=1; one BSX_FUSED_HINT=0; done
EXTRA=--no-witness
for rep in 1 2; do one BSX_FUSED_HINT=1; one BSX_FUSED_HINT=0; done
