"""A GPU load without any kernel of this library: torch matmuls for N seconds."""
import sys, time, torch
secs = float(sys.argv[1])
a = torch.randn(4096, 4096, device='cuda'); b = torch.randn(4096, 4096, device='cuda')
t0 = time.time()
while time.time() - t0 < secs:
    for _ in range(20): c = a @ b
    torch.cuda.synchronize()
