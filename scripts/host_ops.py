"""Which host-side torch ops launch kernels inside one train step (copyBuffer / Fill / elementwise), by call site.

usage: python scripts/host_ops.py [--batch B --classes C]   (on the GPU box)
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--classes", type=int, default=403)
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--batch", str(a.batch), "--classes", str(a.classes)]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from clipfs import synth
    model, tr, cfg = bench.build_trainer(dev, args)
    images = synth.synth_images(args.batch, 224, seed=0).to(dev)
    labels = synth.synth_labels(args.batch, 374, seed=2).to(dev)
    captions = synth.synth_captions(args.classes, 77, cfg.vocab_size, seed=1).to(dev)

    def step():
        tr.flat.zero_grad()
        tr.forward_backward(images, captions, labels, 1, args.batch, row_offset=0)
        tr.optimizer_step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    sites = collections.Counter()

    class Mode(torch.utils._python_dispatch.TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            frames = [f for f in traceback.extract_stack() if "jittor-clip-fewshot_amd" in f.filename or f.filename.endswith("bench.py")]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:][::-1])
            sites[(name, where)] += 1
            return func(*args, **(kwargs or {}))

    with Mode():
        step()
    torch.cuda.synchronize()
    for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
        print(f"{n:4d}  {name:40s} {where}")
    print("total dispatched torch ops in one step:", sum(sites.values()))


if __name__ == "__main__":
    main()
