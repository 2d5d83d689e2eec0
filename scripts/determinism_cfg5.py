"""Run-to-run determinism of whole train steps (a race in a hand-scheduled kernel shows as differing gradients): the same
step (same dropout seed) N times for cfg-5 (ViT-L/14 fp16 mode, 128 images) and cfg-2, flat gradient compared bitwise."""
import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
import bench
from clipfs import synth
dev = torch.device("cuda:0")
for name, mdl, prec, B, C, reps in (("cfg-5", "l14", "fp16", 128, 403, 12), ("cfg-2", "b32", "fp32", 256, 403, 12)):
    args = types.SimpleNamespace(model=mdl, precision=prec, dropout=0.25, trim_text=False, no_shard_text=False, serial_towers=False,
                                 batch=B, classes=C)
    model, tr, cfg = bench.build_trainer(dev, args, model_name=mdl, precision=prec)
    images = synth.synth_images(B, 224, seed=0).to(dev); labels = synth.synth_labels(B, 374, seed=2).to(dev)
    captions = synth.synth_captions(C, 77, cfg.vocab_size, seed=1).to(dev)
    eng = model.engine
    ref = None; bad = 0
    for i in range(reps):
        eng.step = 7  # same dropout seed every time
        tr.flat.zero_grad()
        loss, _, logits = tr.forward_backward(images, captions, labels, 1, B)
        torch.cuda.synchronize()
        gsnap = tr.flat.grads.clone(); lsnap = logits.clone()
        if ref is None:
            ref = (gsnap, lsnap)
        else:
            if not (torch.equal(gsnap, ref[0]) and torch.equal(lsnap, ref[1])):
                bad += 1
                print(f"{name}: run {i} differs: grad max diff {(gsnap - ref[0]).abs().max().item():.3e}, logits {(lsnap - ref[1]).abs().max().item():.3e}", flush=True)
    print(f"{name}: {reps} identical-input steps, {bad} differing; |grad| max {ref[0].abs().max().item():.3e}, finite {bool(torch.isfinite(ref[0]).all())}", flush=True)
    del model, tr
    torch.cuda.empty_cache()
