#!/bin/bash
# Poseidon witness-commitment kernels: kernel trace + SQ counters (separate passes), 256 map-job witnesses
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2prof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktP -o bench -- python tools/poseidon_bench.py 8 > $O/ktP.log 2>&1
python tools/kernel_avg.py $(find $O/ktP -name "bench_kernel_trace.csv" | head -1) > $O/r2_poseidon_kernel_avg.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmcP -o bench -- python tools/poseidon_bench.py 8 > $O/pmcP.log 2>&1
python tools/pmc_summary.py $(find $O/pmcP -name "*counter_collection.csv") > $O/r2_poseidon_pmc_sq.csv
rm -rf $O/ktP $O/pmcP
cat $O/r2_poseidon_kernel_avg.txt; grep -i "leaf_hashes\|merkle_level" $O/r2_poseidon_pmc_sq.csv | cut -d, -f1-5
