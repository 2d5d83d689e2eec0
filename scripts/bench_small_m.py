"""Per-rank (strong scaling, 8 GPUs: 32 images + 51 captions) GEMM shapes under forced tile heights."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
def timeit(f, n=50):
    for _ in range(10): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for Mi, Mt, tag in ((1600, 3927, "8 ranks"), (3200, 7854, "4 ranks"), (6400, 15554, "2 ranks")):
    for name, M, N, K in (("qkv", Mi, 2304, 768), ("out", Mi, 768, 768), ("fc", Mi, 3072, 768), ("proj", Mi, 768, 3072), ("qkv_t", Mi, 768, 2304),
                          ("t_qkv", Mt, 1536, 512), ("t_out", Mt, 512, 512), ("t_fc", Mt, 2048, 512), ("t_proj", Mt, 512, 2048), ("t_qkv_t", Mt, 512, 1536)):
        a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
        t = timeit(lambda: ops.gemm_nt(a, b, out))
        t64 = ((M + 63) // 64) * ((N + 127) // 128)
        print(f"{tag} {name:8s} M={M:5d} N={N:4d} K={K:4d} tiles64={t64:4d}: {t*1e6:7.1f} us {2*M*N*K/t/1e12:6.1f} TF", flush=True)
