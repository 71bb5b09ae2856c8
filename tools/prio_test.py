import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import synth
from blobstreamx_amd.engine import HeaderRangeEngine
J,B,V,R=32,64,100,256
w=synth.Workload(4,R,J,B,v=V)
for prio in (0,-1):
    eng=HeaderRangeEngine(J,B,V,R)
    eng.upload_workload(w)
    main=torch.cuda.Stream(priority=prio)
    with torch.cuda.stream(main):
        for _ in range(3): eng.step()
        torch.cuda.synchronize()
        t=time.perf_counter()
        for _ in range(15): eng.step()
        torch.cuda.synchronize()
        dt=(time.perf_counter()-t)/15
    print("main prio",prio,"ms/step %.2f"%(dt*1e3),"%.1f M/s"%(R*J*B/dt/1e6))
    del eng; torch.cuda.empty_cache()
