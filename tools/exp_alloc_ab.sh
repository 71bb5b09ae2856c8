# headline (single allocation) in fresh processes: allocator x store flavour
one() { env "$@" python bench.py --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']/1e6,1), 'M/s  exp in-region', round(d['roofline']['avg_launch_ms'],3), 'isolated', round(d['roofline']['isolated']['avg_launch_ms'],3), 'subchain', round(d['kernels'][0]['avg_launch_ms'],3))"; }
for i in 1 2 3 4; do one BSX_WITNESS_ALLOC=torch BSX_EXPAND_NT=1; done
for i in 1 2 3 4; do one BSX_WITNESS_ALLOC=vmm BSX_EXPAND_NT=1; done
for i in 1 2; do one BSX_WITNESS_ALLOC=torch BSX_EXPAND_NT=0; done
for i in 1 2; do one BSX_WITNESS_ALLOC=torch BSX_EXPAND_NT=1 BSX_EXPAND_CHUNK=512; done
