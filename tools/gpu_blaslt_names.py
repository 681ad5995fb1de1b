import torch
dev = torch.device("cuda:0")
for (I, J, R) in [(65536, 1536, 384), (65536, 1152, 384), (65536, 384, 1536), (8192, 8192, 8192), (65536, 384, 384)]:
    x = torch.randn(I, R, device=dev).bfloat16(); w = torch.randn(J, R, device=dev).bfloat16()
    for _ in range(3): y = x @ w.t()
torch.cuda.synchronize()
