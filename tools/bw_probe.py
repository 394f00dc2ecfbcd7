"""Device copy bandwidth through torch (read + write), as a yardstick for the image kernels."""
import torch, time
x = torch.empty(1<<30, dtype=torch.uint8, device='cuda'); y = torch.empty_like(x)
for n in (1, 5, 50):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): y.copy_(x)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(n, 'copy GB/s (r+w)', 2*n*(1<<30)/dt/1e9)
