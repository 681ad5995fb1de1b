import torch, time
dev="cuda:0"
for (M,N,K) in [(8192,8192,8192),(65536,1536,384),(65536,384,1536),(65536,1152,384),(4096,4096,4096),(16384,16384,1024)]:
    a=torch.randn(M,K,device=dev).bfloat16(); b=torch.randn(N,K,device=dev).bfloat16()
    for _ in range(3): c=a@b.t()
    torch.cuda.synchronize(); t=time.perf_counter()
    n=20
    for _ in range(n): c=a@b.t()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/n
    print(M,N,K,f"{dt*1e6:.1f} us {2*M*N*K/dt/1e12:.0f} TF")
