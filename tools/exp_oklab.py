import torch, numpy as np, sys
sys.path.insert(0,'.')
import bench, zignal_amd as zg
s=[zg.Image(torch.randint(0,256,(4096,4096,4),dtype=torch.uint8,device='cuda')) for _ in range(4)]
d=[zg.Image(torch.empty((4096,4096,3),dtype=torch.float32,device='cuda')) for _ in range(4)]
t=bench._time_kernel(torch, lambda i: s[i%4].convert(zg.CS_OKLAB, np.float32, out=d[i%4]), n=40, warm=8)
z=[zg.Image(torch.zeros((4096,4096,4),dtype=torch.uint8,device='cuda')) for _ in range(4)]
tz=bench._time_kernel(torch, lambda i: z[i%4].convert(zg.CS_OKLAB, np.float32, out=d[i%4]), n=40, warm=8)
print('oklab 4096^2 rgba8: %.1f us random, %.1f us black' % (t*1e3, tz*1e3))
# reference points for the same traffic shape: a pure write of the destination, and a copy moving the same total bytes
dt=[x.tensor if hasattr(x,'tensor') else None for x in d]
buf=[torch.empty(4096*4096*3, dtype=torch.float32, device='cuda') for _ in range(4)]
tf=bench._time_kernel(torch, lambda i: buf[i%4].fill_(1.0), n=40, warm=8)
a=[torch.empty(4096*4096*2, dtype=torch.float32, device='cuda') for _ in range(4)]
b=[torch.empty(4096*4096*2, dtype=torch.float32, device='cuda') for _ in range(4)]
tc=bench._time_kernel(torch, lambda i: b[i%4].copy_(a[i%4]), n=40, warm=8)
print('fill 201 MB: %.1f us; copy 134 -> 134 MB: %.1f us' % (tf*1e3, tc*1e3))
