"""One launch of each GEMM flavour on the encoder shapes (for rocprofv3 --pmc runs)."""
import sys, torch
sys.path.insert(0, ".")
from dig_amd import ops
dev = torch.device("cuda:0")
I = 65536
def mk(J, R):
    return (torch.randn(I, R, device=dev).bfloat16(), torch.randn(J, R, device=dev).bfloat16())
x, w = mk(1536, 384); y = torch.empty(I, 1536, device=dev, dtype=torch.bfloat16)
for bk in (64, 244):
    for _ in range(3): ops.gemm(x, w, I, 1536, 384, out=y, bk=bk)          # fc1 plain
dy = torch.randn(I, 1536, device=dev).bfloat16(); dx = torch.empty(I, 384, device=dev, dtype=torch.bfloat16)
for bk in (64, 32):
    for _ in range(3): ops.gemm(dy, w, I, 384, 1536, tb=True, out=dx, bk=bk)  # fc1 dgrad
dW = torch.zeros(1536, 384, device=dev)
for _ in range(3): ops.linear_wgrad(dy, x, dW)                               # fc1 wgrad
torch.cuda.synchronize()
