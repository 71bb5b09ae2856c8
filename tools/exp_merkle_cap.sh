# headline vs the resident-workgroup cap of k_header_merkle (162 VGPRs: 3 waves fill a SIMD's register file)
one() { env "$@" python bench.py --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']/1e6,1), 'M/s  exp in-region', round(d['roofline']['avg_launch_ms'],3), 'isolated', round(d['roofline']['isolated']['avg_launch_ms'],3), 'subchain', round(d['kernels'][0]['avg_launch_ms'],3))"; }
for rep in 1 2; do for c in 0 768 512 384 256; do one BSX_MERKLE_WGS=$c; done; done
