import os, time, numpy as np, sys
sys.path.insert(0, os.getcwd())
import oracle, synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "n/a")
print(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0])
w=synth.Workload(4,1,16,64,v=100,mode='S')
vals=w.validators.reshape(-1,100); hh=w.commit_hashes
for th in (1,8,32,64,128,256):
    t=time.perf_counter(); res,ok=oracle.bench_verify_commits(vals,hh,th); dt=time.perf_counter()-t
    print(th, "threads: %.1f k verifies/s"%(ok.size/dt/1e3), flush=True)
