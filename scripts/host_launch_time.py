import os, sys, types, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/jittor-clip-fewshot_amd")
import bench
from clipfs import synth
dev = torch.device("cuda:0")
B, C = int(os.environ.get("B", "32")), int(os.environ.get("C", "51"))
args = types.SimpleNamespace(model="b32", precision="fp32", dropout=0.25, trim_text=False, no_shard_text=False, serial_towers=False, batch=B, classes=C)
model, tr, cfg = bench.build_trainer(dev, args)
images = synth.synth_images(B, 224, seed=0).to(dev); labels = synth.synth_labels(B, 374, seed=2).to(dev)
captions = synth.synth_captions(C, 77, cfg.vocab_size, seed=1).to(dev)
eng = model.engine
host = {}
def wrap(name, fn):
    def inner(*a, **k):
        t0 = time.perf_counter(); out = fn(*a, **k); host.setdefault(name, []).append(time.perf_counter() - t0); return out
    return inner
for n in ("text_forward", "vit_forward", "text_backward", "vit_backward"):
    setattr(eng, n, wrap(n, getattr(eng, n)))
def step():
    tr.flat.zero_grad(); tr.forward_backward(images, captions, labels, 1, B); tr.optimizer_step()
for _ in range(5): step()
torch.cuda.synchronize(); host.clear()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"B={B}: host loop {1e3*(t1-t0)/20:.2f} ms/step (no sync), wall {1e3*(t2-t0)/20:.2f} ms/step")
for k, v in host.items(): print(f"  host time in {k:14s} {1e3*sum(v)/len(v):.3f} ms")
