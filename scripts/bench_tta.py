"""cfg-4 of BASELINE.json: MTA test-time augmentation -- n_img source images x (1 + 64) views through the
ViT-B/32 image tower (LoRA from the shipped checkpoint), L2-normalise, one MTA launch (a workgroup per image),
OOD argmax and top-5.  Views are synthetic and resident in HBM.  Prints one JSON line (images/s, views/s)."""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))


def main():
    import lora_train_vlp as L
    import ood
    from clipfs import ops, synth
    from jclip.model import build_model
    dev = torch.device("cuda:0")
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    V, Cn = 65, 403
    cfg = synth.VIT_B32
    model = build_model(synth.synth_state_dict(cfg, seed=1234), device=dev)
    args = types.SimpleNamespace(encoder="both", position="all", backbone="ViT-B/32", params=["q", "k", "v"], r=4, alpha=1,
                                 dropout_rate=0.25)
    layers = L.apply_lora(args, model)
    L.load_lora(args, layers, os.path.join(ROOT, "tests", "golden", "lora_weights.pkl"))
    model.eval()
    with torch.no_grad():
        text = ops.l2norm_fwd(model.encode_text(synth.synth_captions(Cn, 77, cfg.vocab_size, seed=1).to(dev)))
        views = synth.synth_images(n_img * V, 224, seed=4).reshape(n_img, V, 3, 224, 224).to(dev)

        def step():
            is_base, pred = ood.split_ood(model, views, text)
            logits, _ = ood.mta_scores(model, views, text)
            return ops.topk(logits, 5), is_base

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 5
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f = ops.l2norm_fwd(model.encode_image(views.reshape(n_img * V, 3, 224, 224))).reshape(n_img, V, -1)
        e0.record()
        for _ in range(10):
            ops.mta(f, text)
        e1.record()
        torch.cuda.synchronize()
    # GPU view generation (the reference's CPU/PIL bottleneck): 1 + 64 views of a 500 x 375 image per launch
    import numpy as np
    import tta
    src = torch.from_numpy(np.random.RandomState(0).randint(0, 255, (375, 500, 3), dtype=np.uint8)).to(dev)
    tta.make_tta_views(src, 64)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for s_ in range(20):
        tta.make_tta_views(src, 64, seed=s_)
    torch.cuda.synchronize()
    view_ms = (time.perf_counter() - t1) / 20 * 1e3
    tta.make_tta_views(src, 512)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for s_ in range(5):
        tta.make_tta_views(src, 512, seed=s_)
    torch.cuda.synchronize()
    view512_ms = (time.perf_counter() - t1) / 5 * 1e3
    # each step runs the tower twice (split_ood + mta_scores): report per full pass
    print(json.dumps({"workload": "cfg-4 MTA TTA, ViT-B/32 + LoRA, V=65 views/image, C=403", "n_img": n_img,
                      "images_per_s": round(2 * n_img / dt, 2), "views_per_s": round(2 * n_img * V / dt, 1),
                      "mta_kernel_ms": round(e0.elapsed_time(e1) / 10, 3), "views_65_ms": round(view_ms, 3), "views_513_ms": round(view512_ms, 3), "ms_per_pass": round(dt * 1e3 / 2, 2)}))


if __name__ == "__main__":
    main()
