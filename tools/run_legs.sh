run() { echo "=== $1"; python - <<PY 2>&1 | grep -v amdgpu.ids | tail -3
import sys, json, torch
sys.path.insert(0, ".")
import bench
from blobstreamx_amd import _lib
_lib.lib()
dev = torch.device("cuda:0")
$1
print("OK")
PY
}
run "bench.range_sweep_leg(dev,32,64,100,rs=(1,4,16,64),witness=True); bench.range_sweep_leg(dev,32,64,100,rs=(1,4,16,64),witness=True); bench.range_sweep_leg(dev,32,64,100,rs=(1,4,16,64),witness=True)"
