"""Shader clock / power while the fp32 GEMM (then the f16 GEMM) runs back to back: samples `rocm-smi` once a second from a
child process during ~6 s of launches each.  Evidence for the clock the roofline fractions in DESIGN.md are read against."""
import os, subprocess, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
from clipfs import ops
dev = torch.device("cuda:0")

def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e}"
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "Power", "mclk"))]
    return " | ".join(keep)

print("idle:", sample(), flush=True)
for name, fn in (("fp32 gemm 12800x3072x768", None), ("f16 gemm 8192^3", None)):
    if name.startswith("fp32"):
        a = torch.randn(12800, 768, device=dev); b = torch.randn(3072, 768, device=dev); out = torch.empty(12800, 3072, device=dev)
        run = lambda: ops.gemm_nt(a, b, out)
    else:
        a16 = torch.randn(8192, 8192, device=dev).half(); b = torch.randn(8192, 8192, device=dev); b16 = ops.to_f16(b); out = torch.empty(8192, 8192, device=dev)
        run = lambda: ops.gemm_nt(None, b, out, b_planes=b16, a16=a16)
    t_end = time.time() + 6
    n = 0
    while time.time() < t_end:
        for _ in range(200):
            run()
        n += 200
        if n % 1000 == 0:
            print(name, ":", sample(), flush=True)  # sampled while the queue is full
    torch.cuda.synchronize()
