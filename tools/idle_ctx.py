import sys, time, torch
sys.path.insert(0, '.')
from blobstreamx_amd import _lib
_lib.lib()
torch.cuda.set_device(0)
x = torch.zeros(1 << 20, device='cuda')
ss = [torch.cuda.Stream() for _ in range(16)]
for s in ss:
    with torch.cuda.stream(s): x += 1
torch.cuda.synchronize()
time.sleep(float(sys.argv[1]))
