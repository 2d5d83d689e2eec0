#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): images/sec of the ViT-B/32 forward + LoRA backward train step.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): cfg-2 of BASELINE.json -- ViT-B/32, rank-4 LoRA on q,k,v of all 12+12
blocks (weights = the reference's shipped lora_weights.pkl), 4 learnable text-prompt tokens, one
run_lora step (lora_train_vlp.py:956-1002): 403 captions through the text tower WITH grad, 256 images
through the image tower, 100*cos logits, cross entropy, backward into LoRA + prompt, AdamW.
Synthetic data (N(0,1) images, random captions, CLIP-init random backbone: no network), inputs resident
in HBM before the timed region.  One step = one such pass over one global batch of 256 images.

N > 1 is STRONG scaling (global batch fixed at 256, 256/N images per rank); the
text tower is class-sharded (403/N captions per rank), collectives: all-gather + reduce of the
[403,512] class features and ONE all-reduce of the flat 1.5 MB LoRA+prompt gradient buffer.
``--weak`` keeps 256 images per rank instead.

Rank 0 prints ONE JSON line.  Extra objects: ``roofline`` (dominant kernel = the fp32-MFMA GEMM; every
GEMM launch of one extra, instrumented step is bracketed by HIP events on its launch stream) and
``cpu_baseline`` (the CPU oracle timed on the host cores on a bounded sample; N = 1 only).
In the timed region the text tower runs on a side HIP stream next to the image tower (they are independent
until the logits); the instrumented roofline step serialises them so each GEMM's duration is its own.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# algorithmic GFLOP per unit (SURVEY.md section 8d / BASELINE.md section 2)
IMG_FWD, IMG_BWD = 8.8176 + 0.0221, 8.586 + 0.044
TXT_FWD, TXT_BWD = 5.9595 + 0.0227, 5.959 + 0.045


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="global image batch (strong scaling)")
    ap.add_argument("--classes", type=int, default=403)
    ap.add_argument("--weak", action="store_true", help="keep --batch images PER RANK (weak scaling)")
    ap.add_argument("--no-shard-text", action="store_true", help="replicate the text tower on every rank")
    ap.add_argument("--dropout", type=float, default=0.25, help="LoRA input dropout (reference default 0.25)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--serial-towers", action="store_true",
                    help="run the text and image towers back to back on one stream (default: text tower on a side stream)")
    ap.add_argument("--model", choices=["b32", "l14"], default="b32",
                    help="b32: the headline ViT-B/32 cfg-2; l14: cfg-5 shapes (ViT-L/14, LoRA r=16, synthetic adapters)")
    ap.add_argument("--no-variants", action="store_true", help="skip the opt-in-mode measurements after the timed region")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "fp16"], default="fp32",
                    help="GEMM arithmetic of the towers: exact fp32 MFMA (default) or split-bf16 x3 (opt-in fast mode)")
    ap.add_argument("--trim-text", action="store_true",
                    help="opt-in: skip the text positions after the batch's last EOT (dead under the causal mask); "
                         "NOT the headline configuration")
    ap.add_argument("--forward-only", action="store_true", help="time the zero-grad image forward only (diagnostic)")
    return ap.parse_args()


def build_trainer(dev, args, world):
    import types
    import torch
    import lora_train_vlp as L
    from clipfs import synth
    from jclip.model import build_model
    l14 = args.model == "l14"
    cfg = synth.VIT_L14 if l14 else synth.VIT_B32
    sd = synth.synth_state_dict(cfg, seed=1234)
    model = build_model(sd, device=dev)
    del sd
    largs = types.SimpleNamespace(encoder="both", position="all", backbone="ViT-L/14" if l14 else "ViT-B/32",
                                  params=["q", "k", "v"], r=16 if l14 else 4, alpha=1, dropout_rate=args.dropout)
    layers = L.apply_lora(largs, model)
    if l14:  # no shipped checkpoint for L/14: A ~ U(+-1/sqrt(in)), B ~ N(0, 0.02^2), seed 5
        lw = synth.synth_lora(cfg, 16, seed=5, vision_blocks=range(21))
        names = {"q": "q_proj", "k": "k_proj", "v": "v_proj"}
        with torch.no_grad():
            for i, layer in enumerate(layers):
                for p_ in "qkv":
                    m_ = getattr(layer, names[p_])
                    m_.w_lora_A.copy_(torch.from_numpy(lw[f"layer_{i}"][names[p_]]["w_lora_A"]))
                    m_.w_lora_B.copy_(torch.from_numpy(lw[f"layer_{i}"][names[p_]]["w_lora_B"]))
    else:
        L.load_lora(largs, layers, os.path.join(ROOT, "tests", "golden", "lora_weights.pkl"))
    L.mark_only_lora_as_trainable(model)
    # 4 prompt tokens initialised from the embeddings of "a photo of a" (slow_pace.py:124-131)
    ids = torch.tensor([320, 1125, 539, 320], device=dev)
    ctx = torch.nn.Parameter(model.token_embedding.weight.data[ids].clone())
    model.train()
    tr = L.LoRATrainer(model, prompt_ctx=ctx, shard_text=not args.no_shard_text)
    tr.overlap_towers = not args.serial_towers
    model.engine.trim_text = args.trim_text
    model.engine.precision = args.precision
    return model, tr, cfg


def host_cores() -> int:
    """CPU share of this process: the affinity mask, capped at the GPU box's per-GPU share of 16."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(steps=2):
    """The CPU oracle (oracle/clip_oracle.py, fp32) on a bounded sample of the same workload: 16 images +
    25 captions (the 256:403 ratio), forward + LoRA backward, all host cores."""
    import torch
    from clipfs import safe_pkl, synth
    from oracle import clip_oracle as O
    cfg = synth.VIT_B32
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = synth.synth_state_dict(cfg, seed=1234)
    ck = safe_pkl.load(os.path.join(ROOT, "tests", "golden", "lora_weights.pkl"))
    tl, vl = O.split_lora_checkpoint(ck["weights"], "both", "all", "ViT-B/32", dtype=torch.float32)
    for blk in list(tl.values()) + list(vl.values()):
        for ab in blk.values():
            for t in ab.values():
                t.requires_grad_()
    B, Cn = 8, 13
    img = synth.synth_images(B, 224, seed=0)
    cap = synth.synth_captions(Cn, 77, cfg.vocab_size, seed=1)
    tgt = synth.synth_labels(B, Cn, seed=2)
    ts = []
    t_begin = time.time()
    for i in range(steps + 1):
        t0 = time.time()
        loss, _ = O.train_step_loss(sd, img, cap, tgt, tl, vl, 0.5)
        loss.backward()
        ts.append(time.time() - t0)
        print(f"[bench] cpu baseline step {i}: {ts[-1]:.2f} s", file=sys.stderr, flush=True)
        if time.time() - t_begin > 60 and len(ts) >= 2:
            break
    rest = ts[1:] if len(ts) > 1 else ts
    t = sorted(rest)[len(rest) // 2]
    steps = len(rest)
    return {"value": round(B / t, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 (PyTorch-CPU restatement, NOT Jittor: Jittor is not installable offline), "
                      f"{B} images + {Cn} captions fwd+LoRA-bwd, median of {steps} steps after 1 warm-up"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP engine has no CPU fallback)")
    # CLIPFS_BENCH_REHEARSE=1: run the N-rank code path on a box with fewer GPUs than ranks (ranks share cards, gloo
    # instead of RCCL, which refuses two ranks on one device).  A correctness rehearsal only -- never a measurement.
    rehearse = os.environ.get("CLIPFS_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from clipfs import _lib, dist as D, synth
    model, tr, cfg = build_trainer(dev, args, world)
    gb = args.batch * world if args.weak else args.batch
    lo, hi = D.shard_bounds(gb, rank, world)
    images = synth.synth_images(gb, 224, seed=0)[lo:hi].contiguous().to(dev)
    labels = synth.synth_labels(gb, 374, seed=2)[lo:hi].contiguous().to(dev)
    captions = synth.synth_captions(args.classes, 77, cfg.vocab_size, seed=1).to(dev)
    # caption rows 1..4 are the learnable prompt slots; keep EOT after them (synthetic captions have len >= 6)

    def step():
        if args.forward_only:
            with torch.no_grad():
                model.engine.vit_forward(images, False)
            return
        tr.flat.zero_grad()
        tr.forward_backward(images, captions, labels, 1, gb)
        tr.optimizer_step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        if rank == 0:
            torch.cuda.synchronize()
            print(f"[bench] warmup step {i} done", file=sys.stderr, flush=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    ms = dt / args.steps * 1e3
    value = gb * args.steps / dt

    # ---- roofline leg: one more step with every GEMM launch bracketed by HIP events on its stream ----
    roof = None
    if not args.no_roofline:
        lib = _lib.load()
        overlap = tr.overlap_towers
        tr.overlap_towers = False  # per-kernel durations are only meaningful when nothing else shares the GPU
        lib.clipfs_gemm_timing(1)
        step()
        torch.cuda.synchronize()
        lib.clipfs_gemm_timing(0)
        tr.overlap_towers = overlap
        tms, tfl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        lib.clipfs_gemm_timing_collect(ctypes.byref(tms), ctypes.byref(tfl), ctypes.byref(n))
        if n.value > 0 and tms.value > 0:
            ach = tfl.value / (tms.value * 1e-3) / 1e12
            # achieved counts ALGORITHMIC FLOPs (2MNK); the opt-in 16-bit modes are priced against the dense 16-bit
            # MFMA peak divided by the products they spend per algorithmic multiply (3 for split-bf16, 1 for f16)
            kname, peak = {
                "fp32": ("gemm_nt_kernel<64,128,3> (v_mfma_f32_32x32x2_f32, global_load_lds staging)", FP32_MFMA_PEAK_TFLOPS),
                "bf16x3": ("gemm_bf16x3_kernel<.,.,2> (3 x v_mfma_f32_32x32x16_bf16 per operand pair)", 2500.0 / 3),
                "fp16": ("gemm_f16_kernel<256,128> (v_mfma_f32_32x32x16_f16, both operands f16 via global_load_lds)", 2500.0),
            }[args.precision]
            # HBM traffic per launch from the PMC counters cannot be collected from inside this process: it is the figure
            # of the committed rocprofv3 --pmc passes over this same command (scripts/pmc_traffic.py), exact fp32 kernel only
            traffic, tsrc = None, None
            tpath = os.path.join(ROOT, "profiles", "r01", "gemm_traffic.json")
            if args.precision == "fp32" and args.model == "b32" and os.path.exists(tpath):
                with open(tpath) as f:
                    traffic = json.load(f).get("traffic_bytes_per_launch")
                tsrc = "profiles/r01/gemm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2 per the gfx950 note; L2-side counters: Infinity-Cache hits included)"
            roof = {"bound": "mfma", "kernel": kname,
                    "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": tsrc,
                    "algorithmic_bytes_per_launch": round(lib.clipfs_gemm_timing_last_bytes() / n.value),
                    "launches_per_step": n.value, "avg_launch_us": round(tms.value * 1e3 / n.value, 2),
                    "gflop_per_launch": round(tfl.value / n.value / 1e9, 3),
                    "gemm_ms_per_step": round(tms.value, 3)}
    barrier()

    # ---- opt-in modes next to the headline (same step, same inputs; never the headline `value`) ----
    variants = None
    if world == 1 and not args.no_variants and not args.forward_only and args.model == "b32" and \
            args.precision == "fp32" and not args.trim_text:
        variants = {}
        for name, trim, prec, note in (
                ("trim_text", True, "fp32", "exact fp32; text tower evaluated only up to the last EOT of the batch (positions "
                                            "after a caption's EOT cannot reach its feature under the causal mask)"),
                ("bf16x3", False, "bf16x3", "tower GEMMs as split-bf16 (3 bf16 MFMA products, fp32 accumulate); logits within "
                                            "~3e-4 of fp32, inside the 1e-3 budget"),
                ("bf16x3+trim_text", True, "bf16x3", "both")):
            model.engine.trim_text, model.engine.precision = trim, prec
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            vdt = (time.perf_counter() - t1) / 3
            variants[name] = {"value": round(gb / vdt, 2), "unit": "images/s", "ms_per_step": round(vdt * 1e3, 3), "note": note}
        model.engine.trim_text, model.engine.precision = args.trim_text, args.precision

    if rank == 0:
        n_img_local = hi - lo
        if args.model == "l14":  # 162.03 (+0.303 LoRA) per image, 13.30 per caption; dgrad ~= forward
            step_tflop = (gb * 2 * (162.03 + 0.303) + args.classes * 2 * 13.30) / 1e3
        elif args.forward_only:
            step_tflop = gb * IMG_FWD / 1e3
        else:
            step_tflop = (gb * (IMG_FWD + IMG_BWD) + args.classes * (TXT_FWD + TXT_BWD)) / 1e3
        out = {
            "metric": "images/sec ViT-B/32 fwd+LoRA-bwd bs=256" if args.model == "b32" else
                      "images/sec ViT-L/14 fwd+LoRA-bwd (cfg-5 shapes)", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "f32 (tower GEMMs as split-bf16 x3 MFMA, fp32 accumulate)", "fp16": "f16 operands in the tower GEMMs, fp32 accumulate"}[args.precision], "data": "synthetic",
            "config": {"workload": "cfg-2: ViT-B/32 + rank-4 LoRA(q,k,v; 12 text + 12 vision blocks; shipped "
                                   "lora_weights.pkl) + 4 text-prompt tokens; run_lora train step = text tower "
                                   "fwd+bwd on 403 captions + image tower fwd+bwd + 100*cos CE + AdamW"
                       if not args.forward_only else "ViT-B/32 image tower forward only (diagnostic)",
                       "global_batch": gb, "images_per_rank": n_img_local, "captions": args.classes,
                       "lora_dropout": args.dropout, "tower_streams": 1 if args.serial_towers else 2, "text_positions": "trimmed-to-last-EOT" if args.trim_text else 77, "parallelism": f"dp{world}" + ("" if args.no_shard_text or world == 1 else "+class-sharded-text")},
            "algorithmic_tflop_per_step": round(step_tflop, 3),
            "step_tflops": round(step_tflop / (ms * 1e-3), 2),
            "step_frac_of_fp32_mfma_peak": round(step_tflop / (ms * 1e-3) / (FP32_MFMA_PEAK_TFLOPS * world), 4),
        }
        if roof:
            out["roofline"] = roof
        if variants:
            out["variants"] = variants
        if world == 1 and not args.no_cpu_baseline and not args.forward_only:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
