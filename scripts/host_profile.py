import os, sys, types, time, torch, cProfile, pstats
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/jittor-clip-fewshot_amd")
import bench
from clipfs import synth
dev = torch.device("cuda:0")
B, C = int(os.environ.get("B", "32")), int(os.environ.get("C", "51"))
args = types.SimpleNamespace(model="b32", precision="fp32", dropout=0.25, trim_text=False, no_shard_text=False, serial_towers=False, batch=B, classes=C)
model, tr, cfg = bench.build_trainer(dev, args)
images = synth.synth_images(B, 224, seed=0).to(dev); labels = synth.synth_labels(B, 374, seed=2).to(dev)
captions = synth.synth_captions(C, 77, cfg.vocab_size, seed=1).to(dev)
def step():
    tr.flat.zero_grad(); tr.forward_backward(images, captions, labels, 1, B); tr.optimizer_step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
