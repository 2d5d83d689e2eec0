import torch, time
dev='cuda'
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
for name,M,N,K in [("qkv", 32896, 3072, 1024), ("out", 32896, 1024, 1024), ("fc", 32896, 4096, 1024), ("proj", 32896, 1024, 4096), ("sq8192",8192,8192,8192)]:
    for dt in (torch.float16, torch.bfloat16):
        a=torch.randn(M,K,device=dev,dtype=dt); b=torch.randn(N,K,device=dev,dtype=dt)
        t=timeit(lambda: torch.matmul(a,b.T))
        print(name, dt, f"{t*1e6:.1f} us {2*M*N*K/t/1e12:.1f} TF", flush=True)
