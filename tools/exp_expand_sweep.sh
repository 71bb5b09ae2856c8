# headline vs expansion staging chunk x grid cap (non-temporal, 128-byte aligned stores), fresh process each
one() { env "$@" python bench.py --no-legs 2>/dev/null | python -c "import json,sys; sys.path.insert(0,'.'); from bench_legs.line import detail_of; d=detail_of(sys.stdin.read()); print('$*', round(d['value']/1e6,1), 'M/s  exp in-region', round(d['roofline']['avg_launch_ms'],3), 'isolated', round(d['roofline']['isolated']['avg_launch_ms'],3), 'subchain', round(d['kernels'][0]['avg_launch_ms'],3))"; }
for rep in 1 2 3; do
one BSX_EXPAND_CHUNK=256 BSX_EXPAND_BLOCKS=0
one BSX_EXPAND_CHUNK=512 BSX_EXPAND_BLOCKS=262144
one BSX_EXPAND_CHUNK=512 BSX_EXPAND_BLOCKS=524288
done
