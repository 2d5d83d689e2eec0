"""Which stage of the three-stream cfg-4 pipeline costs wall time?  Variants of the loop of ood.score_stream with stages
removed or moved to the caller's stream.   python scripts/tta_pipeline_probe.py"""
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
    from clipfs import ops, synth
    model.eval()
    rng = np.random.RandomState(7)
    srcs = [torch.from_numpy(rng.randint(0, 256, (375, 500, 3), dtype=np.uint8)).to(dev) for _ in range(8)]
    captions = synth.synth_captions(403, 77, cfg.vocab_size, seed=1).to(dev)
    G = 8
    with torch.no_grad():
        text = ops.l2norm_fwd(model.encode_text(captions))
        bufs = [torch.empty(8, 65, 3, 224, 224, device=dev) for _ in range(2)]
        for i in range(8):
            tta.make_tta_views(srcs[i], 64, seed=i, out=bufs[0][i])
            tta.make_tta_views(srcs[i], 64, seed=i, out=bufs[1][i])
        main_s = torch.cuda.current_stream(dev)
        s1, s2 = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=-1)

        def run(views_on, mta_on):
            """views_on / mta_on: None (skip), 'main', 'side'"""
            consumed = [None, None]
            evs = [None, None]

            def gen(g):
                st = s1 if views_on == "side" else main_s
                with torch.cuda.stream(st):
                    if views_on == "side" and consumed[g % 2] is not None:
                        st.wait_event(consumed[g % 2])
                    for i in range(8):
                        tta.make_tta_views(srcs[i], 64, seed=g * 8 + i, out=bufs[g % 2][i])
                    e = torch.cuda.Event()
                    e.record(st)
                    evs[g % 2] = e
            if views_on:
                gen(0)
            for g in range(G):
                if views_on == "side":
                    main_s.wait_event(evs[g % 2])
                f = model.encode_image(bufs[g % 2].reshape(520, 3, 224, 224))
                d = torch.cuda.Event()
                d.record(main_s)
                consumed[g % 2] = d
                if views_on and g + 1 < G:
                    gen(g + 1)
                f = ops.l2norm_fwd(f.contiguous())
                if mta_on == "main":
                    ops.topk(ops.mta(f.reshape(8, 65, -1), text)[1], 5)
                elif mta_on == "side":
                    e = torch.cuda.Event()
                    e.record(main_s)
                    with torch.cuda.stream(s2):
                        s2.wait_event(e)
                        f.record_stream(s2)
                        ops.topk(ops.mta(f.reshape(8, 65, -1), text)[1], 5)
            main_s.wait_stream(s1)
            main_s.wait_stream(s2)

        for name, v, m in (("tower only", None, None), ("views main", "main", None), ("views side", "side", None),
                           ("mta main", None, "main"), ("mta side", None, "side"), ("both main", "main", "main"),
                           ("both side", "side", "side")):
            run(v, m)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(v, m)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"{name:12s} {(t2 - t0) / G * 1e3:7.2f} ms per group (host {(t1 - t0) / G * 1e3:6.2f})", flush=True)


if __name__ == "__main__":
    main()
