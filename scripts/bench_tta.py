"""cfg-4 of BASELINE.json: MTA test-time augmentation -- n_img source images x (1 + 64) views through the
ViT-B/32 image tower (LoRA from the shipped checkpoint), L2-normalise, one MTA launch (a workgroup per image),
OOD argmax and top-5.  Views are synthetic and resident in HBM.  Prints one JSON line (images/s, views/s).

    python scripts/bench_tta.py 32                       # one GPU, 32 source images
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29501 \
        scripts/bench_tta.py 32                          # 8 GPUs, 32 images per rank: images sharded, results gathered
"""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))


def main():
    import torch.distributed as dist
    import lora_train_vlp as L
    import ood
    from clipfs import ops, synth
    from jclip.model import build_model
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = os.environ.get("CLIPFS_BENCH_REHEARSE") == "1"  # ranks share cards, gloo: correctness rehearsal only
    if rehearse:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 8   # source images PER RANK (weak scaling: images are independent units)
    total = n_img * world
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
        # this rank's images (shard source images, every image's views stay on one GPU)
        views = synth.synth_images(n_img * V, 224, seed=4 + rank).reshape(n_img, V, 3, 224, 224).to(dev)
        lo0 = rank * n_img

        def step():
            # one tower pass -> MTA -> top-5 + base/new; the integer table is assembled on every rank
            return ood.score_images_sharded(lambda lo, hi: ood.score_views(model, views[lo - lo0:hi - lo0], text)[:2], total, dev)

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(2):
            step()
        barrier()
        t0 = time.perf_counter()
        iters = 5
        for _ in range(iters):
            top5, is_base = step()
        barrier()
        dt = (time.perf_counter() - t0) / iters
        if world > 1:
            tmax = torch.tensor([dt], device=dev if not rehearse else "cpu", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = tmax.item()
        assert top5.shape == (total, 5) and is_base.shape == (total,)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f = ops.l2norm_fwd(model.encode_image(views.reshape(n_img * V, 3, 224, 224))).reshape(n_img, V, -1)
        e0.record()
        for _ in range(10):
            ops.mta(f, text)
        e1.record()
        torch.cuda.synchronize()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
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
    print(json.dumps({"workload": "cfg-4 MTA TTA, ViT-B/32 + LoRA, V=65 views/image, C=403: tower pass + MTA + top-5 / OOD split",
                      "n_gpus": world, "scaling": "weak", "images_per_rank": n_img, "images_per_s": round(total / dt, 2),
                      "views_per_s": round(total * V / dt, 1), "ms_per_pass": round(dt * 1e3, 2),
                      "mta_kernel_ms": round(e0.elapsed_time(e1) / 10, 3), "views_65_ms": round(view_ms, 3),
                      "views_513_ms": round(view512_ms, 3)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
