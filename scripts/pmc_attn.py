"""A few launches of the fp32 MFMA attention kernels at the path's shapes (for rocprofv3 --pmc passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
for B, L, H, causal in ((256, 50, 12, False), (403, 77, 8, True)):
    qkv = torch.randn(B * L, 3 * H * 64, device=dev); do = torch.randn(B * L, H * 64, device=dev)
    for _ in range(3):
        o, l = ops.attention_fwd(qkv, B, L, H, causal, want_lse=True)
        ops.attention_bwd(qkv, do, B, L, H, causal, out=o, lse=l)
    torch.cuda.synchronize()
