for leg in "bench.latency_leg(dev,32,64,100)" "bench.concurrent_leg(dev,32,64,100)" "bench.range_sweep_leg(dev,32,64,100)" "bench.range_sweep_leg(dev,32,64,100,rs=(1,4,16,64),witness=True)" "bench.commitment_leg(dev,32,64,100,bench.calibrate(dev))"; do
echo "=== $leg"
python - <<PY 2>&1 | tail -4
import sys, json, torch
sys.path.insert(0, ".")
import bench
from blobstreamx_amd import _lib
_lib.lib()
dev = torch.device("cuda:0")
r = $leg
print(json.dumps(r)[:1500])
PY
done
