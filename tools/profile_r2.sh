#!/bin/bash
# Round-2 profile set (run on the GPU box: gpurun -- 'bash tools/profile_r2.sh').  Counter passes are separate rocprofv3
# runs with --kernel-trace only (guide: no --pmc together with sys/hip traces).  Outputs under gpurun_out/r2prof/, the
# summaries are copied to profiles/ by hand.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2prof; mkdir -p $O
HEAD="python bench.py --no-legs --steps 10 --warmup 2"
# 1. the bench line itself (all legs)
python bench.py > $O/r2_bench_n1.json 2> $O/bench.err
# 2. kernel trace + stats of the headline command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- $HEAD > $O/kt.log 2>&1
python tools/kernel_avg.py $(find $O/kt -name "bench_kernel_trace.csv" | head -1) > $O/r2_kernel_avg_steady_state.txt
cp $(find $O/kt -name "bench_kernel_stats.csv" | head -1) $O/r2_rocprofv3_kernel_stats.csv
# 3. HBM traffic PMC passes (separate runs)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o bench -- python bench.py --no-legs --steps 3 --warmup 1 > $O/pmc_$c.log 2>&1
done
python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*counter_collection.csv") > $O/r2_pmc_hbm_traffic.csv
echo '{"jobs_per_launch": 4096, "batch": 64, "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --no-legs --steps 3 --warmup 1"}' > $O/r2_pmc_hbm_traffic.meta.json
# 4. header_range_1024 (BASELINE config #3: rocprof counters on the 1024 shape)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1024 -o bench -- $HEAD --batch 32 > $O/kt1024.log 2>&1
python tools/kernel_avg.py $(find $O/kt1024 -name "bench_kernel_trace.csv" | head -1) > $O/r2_1024_kernel_avg_steady_state.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc1024_$c -o bench -- python bench.py --no-legs --steps 3 --warmup 1 --batch 32 > $O/pmc1024_$c.log 2>&1
done
python tools/pmc_summary.py $(find $O/pmc1024_FETCH_SIZE $O/pmc1024_WRITE_SIZE -name "*counter_collection.csv") > $O/r2_1024_pmc_hbm_traffic.csv
# 5. mode S (2048 x 512): kernel trace + SQ counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktS -o bench -- python bench.py --mode S --validators 512 --cpu-seconds 1 > $O/ktS.log 2>&1
python tools/kernel_avg.py $(find $O/ktS -name "bench_kernel_trace.csv" | head -1) > $O/r2_modeS_512_kernel_avg.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmcS -o bench -- python bench.py --mode S --validators 512 --cpu-seconds 1 > $O/pmcS.log 2>&1
python tools/pmc_summary.py $(find $O/pmcS -name "*counter_collection.csv") > $O/r2_modeS_512_pmc_sq.csv
# 6. SQ counters of the headline + the Poseidon leg
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmcSQ -o bench -- python bench.py --no-legs --steps 3 --warmup 1 > $O/pmcSQ.log 2>&1
python tools/pmc_summary.py $(find $O/pmcSQ -name "*counter_collection.csv") > $O/r2_pmc_sq.csv
# 7. ceilings
./tools/microbench_alu > $O/r2_microbench_alu.txt 2>&1
./tools/microbench > $O/r2_microbench.txt 2>&1
rm -rf $O/kt $O/kt1024 $O/ktS $O/pmc_* $O/pmc1024_* $O/pmcS $O/pmcSQ     # raw traces stay on the box (size)
# 8. the other profile scripts: Poseidon kernels, compact-only pipeline, one host-tier call, headline timeline
for s in profile_r2_poseidon profile_r2_compact profile_latency profile_headline_timeline; do bash tools/$s.sh > $O/$s.log 2>&1; done
ls -la $O
