"""Host time of the cfg-4 stages (enqueue only, GPU idle before and synchronised after each): is the pipeline of
ood.score_stream host-bound?   python scripts/tta_host_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model, tr, cfg = bench.build_trainer(dev, args)
    import tta
    from clipfs import ops, synth, views
    model.eval()
    rng = np.random.RandomState(7)
    srcs = [torch.from_numpy(rng.randint(0, 256, (375, 500, 3), dtype=np.uint8)).to(dev) for _ in range(8)]
    captions = synth.synth_captions(403, 77, cfg.vocab_size, seed=1).to(dev)
    with torch.no_grad():
        text = ops.l2norm_fwd(model.encode_text(captions))

        def stage(name, fn, n=5):
            fn()
            torch.cuda.synchronize()
            host = gpu = 0.0
            for _ in range(n):
                t0 = time.perf_counter()
                out = fn()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                host += t1 - t0
                gpu += t2 - t0
            print(f"{name:28s} host enqueue {host / n * 1e3:7.2f} ms   until done {gpu / n * 1e3:7.2f} ms", flush=True)
            return out

        stage("view_records x8 (numpy)", lambda: [views.view_records(500, 375, 64, seed=i) for i in range(8)])
        vs = stage("make_tta_views x8", lambda: [tta.make_tta_views(srcs[i], 64, seed=i) for i in range(8)])
        v = stage("stack", lambda: torch.stack(vs))
        f = stage("encode_image 520 views", lambda: model.encode_image(v.reshape(520, 3, 224, 224)))
        fn = stage("l2norm", lambda: ops.l2norm_fwd(f.contiguous()))
        stage("mta + topk", lambda: ops.topk(ops.mta(fn.reshape(8, 65, -1), text)[1], 5))

        def loop(name, fn, n=6):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"{name:44s} {(t2 - t0) / n * 1e3:7.2f} ms per pass (host issue {(t1 - t0) / n * 1e3:6.2f})", flush=True)

        x = v.reshape(520, 3, 224, 224)
        loop("tower only", lambda: model.encode_image(x))
        loop("tower + l2norm + mta + topk", lambda: ops.topk(ops.mta(ops.l2norm_fwd(model.encode_image(x).contiguous()).reshape(8, 65, -1), text)[1], 5))
        loop("views + stack only", lambda: torch.stack([tta.make_tta_views(srcs[i], 64, seed=i) for i in range(8)]))
        import ood
        loop("score_views on fresh views (one stream)", lambda: ood.score_views(model, torch.stack([tta.make_tta_views(srcs[i], 64, seed=i) for i in range(8)]), text))
        loop("score_stream, 6 groups per call", lambda: ood.score_stream(model, srcs * 6, text, n_crops=64, images_per_pass=8), n=2)


if __name__ == "__main__":
    main()
