"""When does each tower finish inside the train step?  Events on the two tower streams at the end of the forward and the
backward halves of LoRATrainer.forward_backward (cfg-2 shapes), 10 steps after 3 warm-ups."""
import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
import bench
from clipfs import synth, ops
dev = torch.device("cuda:0")
args = types.SimpleNamespace(model="b32", precision="fp32", dropout=0.25, trim_text=False, no_shard_text=False, serial_towers=False,
                             batch=int(os.environ.get("B", "256")), classes=int(os.environ.get("C", "403")))
model, tr, cfg = bench.build_trainer(dev, args)
B, C = args.batch, args.classes
images = synth.synth_images(B, 224, seed=0).to(dev); labels = synth.synth_labels(B, 374, seed=2).to(dev)
captions = synth.synth_captions(C, 77, cfg.vocab_size, seed=1).to(dev)
eng = model.engine
marks = {}
def wrap(name, fn):
    def inner(*a, **k):
        out = fn(*a, **k)
        e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream()); marks.setdefault(name, []).append(e)
        return out
    return inner
eng.text_forward = wrap("text_fwd_end", eng.text_forward); eng.vit_forward = wrap("img_fwd_end", eng.vit_forward)
eng.text_backward = wrap("text_bwd_end", eng.text_backward); eng.vit_backward = wrap("img_bwd_end", eng.vit_backward)
def step():
    e0 = torch.cuda.Event(enable_timing=True); e0.record(); marks.setdefault("start", []).append(e0)
    tr.flat.zero_grad(); tr.forward_backward(images, captions, labels, 1, B); tr.optimizer_step()
    e1 = torch.cuda.Event(enable_timing=True); e1.record(); marks.setdefault("end", []).append(e1)
for _ in range(3): step()
torch.cuda.synchronize(); marks.clear()
for _ in range(10): step()
torch.cuda.synchronize()
n = len(marks["start"])
for k in ("text_fwd_end", "img_fwd_end", "text_bwd_end", "img_bwd_end", "end"):
    v = [marks["start"][i].elapsed_time(marks[k][i]) for i in range(n)]
    print(f"{k:14s} {sum(v)/n:8.2f} ms after step start", flush=True)
