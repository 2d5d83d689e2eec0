"""Attention kernels at the cfg-2 shapes (image: 256 x 12 heads x 50 tokens; text: 403 x 8 heads x 77 tokens, causal):
run under rocprofv3 --kernel-trace --stats to get per-kernel times (fwd, dQ pass, dK/dV pass)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")
for B, L, H, causal in ((256, 50, 12, False), (403, 77, 8, True)):
    qkv = torch.randn(B * L, 3 * H * 64, device=dev)
    do = torch.randn(B * L, H * 64, device=dev)
    o_, l_ = ops.attention_fwd(qkv, B, L, H, causal, want_lse=True)
    for _ in range(10):
        ops.attention_fwd(qkv, B, L, H, causal, want_lse=True)
        ops.attention_bwd(qkv, do, B, L, H, causal, out=o_, lse=l_)
torch.cuda.synchronize()
