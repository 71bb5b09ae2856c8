#!/bin/bash
# tools: VERDICT r5 #6a — the Poseidon leaf kernel at 5 / 6 waves per SIMD: the S-box asm block's fixed temporaries moved down
# (v[66:95] / v[50:79]) and __launch_bounds__(256, 5 / 6).  Builds libbsx_p5.so / libbsx_p6.so beside the product library (never
# shipped); run tools/poseidon_bench.py with BSX_LIB_OVERRIDE on the GPU box.  usage (CPU box): tools/exp_poseidon_occ.sh
set -e
cd "$(dirname "$0")/../blobstreamx_amd/csrc"
for cfg in "5 66" "6 50"; do
  set -- $cfg; W=$1; BASE=$2
  mkdir -p build_p$W/inc
  BSX_SBOX_BASE=$BASE BSX_SBOX_OUT=$PWD/build_p$W/inc/goldilocks_sbox_asm.h python ../../tools/gen_gl_sbox_asm.py
  # the variant header shadows the committed one for kernels_poseidon.hip only (it is the only user of the block)
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed -DBSX_LEAF_WAVES=$W -include hip/hip_runtime.h -include build_p$W/inc/goldilocks_sbox_asm.h \
        -Rpass-analysis=kernel-resource-usage -c kernels_poseidon.hip -o build_p$W/kernels_poseidon.o 2> build_p$W/resource.txt || { tail -20 build_p$W/resource.txt; exit 1; }
  grep -A12 "k_leaf_hashesILb1" build_p$W/resource.txt | grep -E "Function Name|VGPRs:|Spill|Occupancy" | head -8
  OBJS=$(ls build/*.o | grep -v kernels_poseidon.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libbsx_p$W.so $OBJS build_p$W/kernels_poseidon.o
done
