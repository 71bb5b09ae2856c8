#!/bin/bash
# Round-6 profile set (run on the GPU box: gpurun -- 'bash tools/profile_r6.sh').  Counter passes are separate rocprofv3
# runs with --kernel-trace only (guide: no --pmc together with sys/hip traces).  Outputs under gpurun_out/r6prof/; the
# summaries are copied to profiles/ afterwards.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6prof; mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
HEAD="python bench.py --no-legs --no-autotune --steps 10 --warmup 2 --long-steps 0"   # traced runs keep the default stream assignment: the autotune trials would sit in the trace
kt() { find $1 -name "*kernel_trace.csv" | head -1; }
# 1. kernel trace + stats of the headline command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- $HEAD > $O/kt.log 2>&1
python tools/kernel_avg.py $(kt $O/kt) > $O/r6_kernel_avg_steady_state.txt
cp $(find $O/kt -name "bench_kernel_stats.csv" | head -1) $O/r6_rocprofv3_kernel_stats.csv
python tools/timeline2.py $(kt $O/kt) > $O/r6_timeline_steady_state.txt
# 2. HBM traffic PMC passes (separate runs)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o bench -- python bench.py --no-legs --no-autotune --steps 3 --warmup 1 --long-steps 0 > $O/pmc_$c.log 2>&1
done
python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*counter_collection.csv") > $O/r6_pmc_hbm_traffic.csv
echo '{"jobs_per_launch": 4096, "batch": 64, "layout": "map", "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --no-legs --no-autotune --steps 3 --warmup 1"}' > $O/r6_pmc_hbm_traffic.meta.json
# 2b. BASELINE config #3: header_range_1024 (32 x 32) — kernel averages + the HBM counters of ITS expansion launches
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1024 -o bench -- $HEAD --batch 32 > $O/kt1024.log 2>&1
python tools/kernel_avg.py $(kt $O/kt1024) > $O/r6_1024_kernel_avg_steady_state.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc1024_$c -o bench -- python bench.py --batch 32 --no-legs --no-autotune --steps 3 --warmup 1 --long-steps 0 > $O/pmc1024_$c.log 2>&1
done
python tools/pmc_summary.py $(find $O/pmc1024_FETCH_SIZE $O/pmc1024_WRITE_SIZE -name "*counter_collection.csv") > $O/r6_1024_pmc_hbm_traffic.csv
echo '{"jobs_per_launch": 4096, "batch": 32, "layout": "map", "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --batch 32 --no-legs --no-autotune --steps 3 --warmup 1"}' > $O/r6_1024_pmc_hbm_traffic.meta.json
# 2c. mode S, V = 100: the COMMIT units' expansion (2048 units per launch)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcSU_$c -o bench -- python bench.py --mode S --validators 100 --cpu-seconds 1 > $O/pmcSU_$c.log 2>&1
done
python tools/pmc_summary.py $(find $O/pmcSU_FETCH_SIZE $O/pmcSU_WRITE_SIZE $O/pmcSU5_FETCH_SIZE $O/pmcSU5_WRITE_SIZE -name "*counter_collection.csv") > $O/r6_modeS_100_units_pmc_hbm_traffic.csv
echo '{"units_per_launch": 2048, "layout": "commit", "v": 100, "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --mode S --validators 100 --cpu-seconds 1"}' > $O/r6_modeS_100_units_pmc_hbm_traffic.meta.json
# 2d. mode S, V = 512: the COMMIT units' expansion (stress.v512.witness.roofline.traffic was null)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcSU5_$c -o bench -- python bench.py --mode S --validators 512 --cpu-seconds 1 > $O/pmcSU5_$c.log 2>&1
done
python tools/pmc_summary.py $(find $O/pmcSU5_FETCH_SIZE $O/pmcSU5_WRITE_SIZE -name "*counter_collection.csv") > $O/r6_modeS_512_units_pmc_hbm_traffic.csv
echo '{"units_per_launch": 2048, "layout": "commit", "v": 512, "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --mode S --validators 512 --cpu-seconds 1"}' > $O/r6_modeS_512_units_pmc_hbm_traffic.meta.json
# 3. SQ counters of the headline
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmcSQ -o bench -- python bench.py --no-legs --no-autotune --steps 3 --warmup 1 --long-steps 0 > $O/pmcSQ.log 2>&1
python tools/pmc_summary.py $(find $O/pmcSQ -name "*counter_collection.csv") > $O/r6_pmc_sq.csv
# 4. mode S (2048 x 512 and 2048 x 100): kernel trace + SQ counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktS -o bench -- python bench.py --mode S --validators 512 --cpu-seconds 1 > $O/ktS.log 2>&1
python tools/kernel_avg.py $(kt $O/ktS) > $O/r6_modeS_512_kernel_avg.txt
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmcS -o bench -- python bench.py --mode S --validators 512 --cpu-seconds 1 > $O/pmcS.log 2>&1
python tools/pmc_summary.py $(find $O/pmcS -name "*counter_collection.csv") > $O/r6_modeS_512_pmc_sq.csv
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmcS1 -o bench -- python bench.py --mode S --validators 100 --cpu-seconds 1 > $O/pmcS1.log 2>&1
python tools/pmc_summary.py $(find $O/pmcS1 -name "*counter_collection.csv") > $O/r6_modeS_100_pmc_sq.csv
# 5. Poseidon commitment kernels (256 map-job witnesses): kernel trace + SQ counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktP -o bench -- python tools/poseidon_bench.py 8 > $O/ktP.log 2>&1
python tools/kernel_avg.py $(kt $O/ktP) > $O/r6_poseidon_kernel_avg.txt
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmcP -o bench -- python tools/poseidon_bench.py 8 > $O/pmcP.log 2>&1
python tools/pmc_summary.py $(find $O/pmcP -name "*counter_collection.csv") > $O/r6_poseidon_pmc_sq.csv
# 6. compact-only pipeline (2 buffer sets): kernel averages + timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktC -o bench -- python bench.py --no-legs --no-autotune --no-witness --engines 1 --alternate 2 --steps 12 --warmup 2 --long-steps 0 > $O/ktC.log 2>&1
python tools/kernel_avg.py $(kt $O/ktC) > $O/r6_compact_kernel_avg_steady_state.txt
python tools/timeline3.py $(kt $O/ktC) 6 2 > $O/r6_compact_timeline.txt
# 7. one host-tier call (bsx_header_range)
rocprofv3 --kernel-trace --output-format csv -d $O/ktL -o lat -- python tools/latency_probe.py 30 > $O/ktL.log 2>&1
python tools/timeline3.py $(kt $O/ktL) 1.1 0 > $O/r6_latency_timeline.txt
# 7b. mode S, one step in flight: the kernels of a joined step
rocprofv3 --kernel-trace --output-format csv -d $O/ktJ -o ms -- python tools/modeS_joined_trace.py 100 12 > $O/ktJ.log 2>&1
python tools/timeline3.py $(kt $O/ktJ) 2.0 0 > $O/r6_modeS_100_joined_timeline.txt
rm -rf $O/ktJ
# 8. commit-check stages alone
python tools/commit_chain_time.py > $O/r6_commit_chain_stages.txt 2>&1
# 9. VALU instructions per unit of work (for bench.py's valu_issue_frac): fused leaf hashing 256 jobs x 3332 rows x 17 permutations;
#    keyed verification of 1,048,576 signatures (mode S 512)
python tools/valu_insts.py $O/valu_insts.json "poseidon_leaf=$O/r6_poseidon_pmc_sq.csv:k_leaf_hashes<true>:14500881:Poseidon permutation" \
   "ed25519_keyed_1lane=$O/r6_modeS_512_pmc_sq.csv:k_ed25519_verify_keyed<true, true, 1>:1048576:signature" \
   "ed25519_keyed_mixed=$O/r6_modeS_100_pmc_sq.csv:k_ed25519_verify_keyed_mixed:204800:signature" > $O/valu_insts.log 2>&1
rm -rf $O/kt $O/kt1024 $O/pmc1024_FETCH_SIZE $O/pmc1024_WRITE_SIZE $O/pmcSU_FETCH_SIZE $O/pmcSU_WRITE_SIZE $O/pmcSU5_FETCH_SIZE $O/pmcSU5_WRITE_SIZE $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmcSQ $O/ktS $O/pmcS $O/pmcS1 $O/ktP $O/pmcP $O/ktC $O/ktL
# 10. the bench line itself (all legs) — after valu_insts.json exists so that valu_issue_frac is filled in
cp $O/valu_insts.json profiles/valu_insts.json
# the traffic figures of this run's own counter passes (bench.py reads profiles/*pmc_hbm_traffic*.csv + .meta.json at run time)
cp $O/r6_pmc_hbm_traffic.csv $O/r6_pmc_hbm_traffic.meta.json $O/r6_1024_pmc_hbm_traffic.csv $O/r6_1024_pmc_hbm_traffic.meta.json \
   $O/r6_modeS_100_units_pmc_hbm_traffic.csv $O/r6_modeS_100_units_pmc_hbm_traffic.meta.json \
   $O/r6_modeS_512_units_pmc_hbm_traffic.csv $O/r6_modeS_512_units_pmc_hbm_traffic.meta.json profiles/
python bench.py > $O/r6_bench_n1.out 2> $O/bench.err
# the two stdout lines apart: the full object (DETAIL) and the compact line the driver parses
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
from bench_legs.line import detail_of
txt = open("gpurun_out/r6prof/r6_bench_n1.out").read()
json.dump(detail_of(txt), open("gpurun_out/r6prof/r6_bench_n1.json", "w"))
open("gpurun_out/r6prof/r6_bench_n1_compact_line.json", "w").write(txt.rstrip().splitlines()[-1] + "\n")
PY
# 11. the driver's own form of the command (its record is BENCH_r06.json): 20 timed steps, 5 warm-up
python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_driver_form.err | tail -n 1 > $O/r6_bench_n1_driver_form_compact_line.json
# 12. the lone-caller serial path of the coalescing front end, A/B (the round-5 claim had no profile)
python tools/lone_caller_ab.py > $O/r6_lone_caller_ab.txt 2>> $O/bench.err
[ -f blobstreamx_amd/lib/libbsx_nofast.so ] && BSX_LIB_OVERRIDE=$PWD/blobstreamx_amd/lib/libbsx_nofast.so python tools/lone_caller_ab.py >> $O/r6_lone_caller_ab.txt 2>> $O/bench.err
ls -la $O
