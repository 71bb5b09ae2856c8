# expansion-launch time over 8 hipMalloc'ed buffers for several store/staging variants of k_expand_witness (one process each)
run() { echo "== $*"; env "$@" python tools/exp_alloc2.py torch torch torch torch torch torch torch torch 2>/dev/null | head -8 | awk '{printf "%s ", $5} END {print ""}'; }
run BSX_EXPAND_CHUNK=256 BSX_EXPAND_NT=1
run BSX_EXPAND_CHUNK=256 BSX_EXPAND_NT=0
run BSX_EXPAND_CHUNK=512 BSX_EXPAND_NT=1
run BSX_EXPAND_CHUNK=2048 BSX_EXPAND_NT=0 BSX_EXPAND_BLOCKS=16384
run BSX_EXPAND_CHUNK=1024 BSX_EXPAND_NT=1 BSX_EXPAND_BLOCKS=65536
