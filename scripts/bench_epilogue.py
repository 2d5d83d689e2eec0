"""GEMM epilogue cost at the path's shapes: plain vs bias vs bias+LoRA vs QuickGELU(+aux) vs residual."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for name, M, N, K in (("qkv", 12800, 2304, 768), ("fc", 12800, 3072, 768), ("proj", 12800, 768, 3072), ("t_qkv", 31031, 1536, 512)):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) * K ** -0.5; out = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); aux = torch.empty(M, N, device=dev)
    t = torch.randn(M, 12, device=dev); lb = torch.randn(N, 4, device=dev)
    for _ in range(20): ops.gemm_nt(a, b, out)  # clock ramp
    r = {}
    r["plain"] = timeit(lambda: ops.gemm_nt(a, b, out))
    r["bias"] = timeit(lambda: ops.gemm_nt(a, b, out, bias=bias))
    if N % 3 == 0:
        r["bias+lora"] = timeit(lambda: ops.gemm_nt(a, b, out, bias=bias, lora_t=t, lora_b=lb, lora_seg_width=N // 3, lora_scale=0.5))
    r["bias+gelu+aux"] = timeit(lambda: ops.gemm_nt(a, b, out, bias=bias, act=1, aux_out=aux))
    r["bias+res"] = timeit(lambda: ops.gemm_nt(a, b, out, bias=bias, residual=res))
    r["x gelu'(aux)"] = timeit(lambda: ops.gemm_nt(a, b, out, act=2, aux_in=aux))
    print(name, "  ".join(f"{k} {v*1e6:.1f}us" for k, v in r.items()), flush=True)
