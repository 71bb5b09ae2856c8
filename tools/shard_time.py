#!/usr/bin/env python3
"""Per-rank cost of the N-GPU configuration on ONE GPU: rank 0's engine of a world-N run (its J/N-job slice of all N*R
ranges, local fold, top fold of its R ranges, commit check, expansion) with the all-gather replaced by a device-side
copy of this rank's own partial records into every rank slot.  Weak scaling holds if this matches the N = 1 step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, synth
import blobstreamx_amd.engine as E
J, B, V, R = 32, 64, 100, int(os.environ.get("R", "256"))
for world in [int(x) for x in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    def fake_gather(partial, w_, rt, out, async_op=False):
        flat = partial[:rt * 128]
        g = out[:w_ * rt * 128].view(w_, rt * 128)
        g.copy_(flat.unsqueeze(0).expand(w_, rt * 128))      # every "rank" contributes this rank's records: shapes and traffic as in the real gather
        return (out[:w_ * rt * 128], None) if async_op else out[:w_ * rt * 128]
    E.all_gather_records = fake_gather
    w = synth.Workload(4, R * world, J, B, v=V)
    eng = E.PipelinedEngines(J, B, V, R, n_engines=2, rank=0, world=world)
    eng.upload_workload(w)
    for _ in range(4): eng.step()
    eng.join(); torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N): eng.step()
    eng.join(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print("world %d: rank-0 step %.3f ms  (%d ranges x %d jobs per rank) -> %.1f M headers/s per GPU" % (world, dt * 1e3, R * world, J // world, R * J * B / dt / 1e6), flush=True)
    del eng, w
    torch.cuda.empty_cache()
