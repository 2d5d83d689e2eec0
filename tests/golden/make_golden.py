#!/usr/bin/env python3
"""Generates the committed golden vectors from the CPU oracle (fp64 truth, stored as fp64/fp32 .npz).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The reference ships NO golden vectors and cannot run here (Jittor is not installable offline, SURVEY.md
section 8c), so these pin the build's own restatement: tests/test_oracle_golden.py fails when the oracle
drifts, tests/test_golden_gpu.py checks the HIP engine against the same files on the GPU box (where
/root/reference does not exist).  Inputs are regenerated from seeds by clipfs.synth; only outputs (and
small inputs that are not seed-derived) are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))
sys.path.insert(0, ROOT)

from clipfs import safe_pkl, synth  # noqa: E402
from oracle import clip_oracle as O  # noqa: E402

SENTENCES = ["a diagram", "a dog", "a cat", "a photo of a", "A photo of a Bear, a type of animal.",
             "it's  5 o'clock!!  café", "a photo of a stop sign.", "<|startoftext|>hello<|endoftext|>",
             "Thu-vien_quoc-gia 12345 &amp; co."]


def tiny_case():
    """TINY CLIP + rank-4 LoRA(q,k,v,o) + 4 prompt tokens: logits, loss, every gradient."""
    cfg = synth.TINY
    sd = {k: v.double() for k, v in synth.synth_state_dict(cfg, seed=21, perturb=True).items()}
    lw = synth.synth_lora(cfg, 4, seed=22, params=("q", "k", "v", "o"))
    nt = cfg.transformer_layers
    conv = lambda d: {p: {k: torch.from_numpy(v).double().requires_grad_() for k, v in ab.items()} for p, ab in d.items()}
    tl = {b: conv(lw[f"layer_{b}"]) for b in range(nt)}
    vl = {b: conv(lw[f"layer_{nt + b}"]) for b in range(cfg.vision_layers)}
    B, Cn = 5, 7
    img = synth.synth_images(B, cfg.image_resolution, seed=23).double()
    cap = synth.synth_captions(Cn, cfg.context_length, cfg.vocab_size, seed=24, max_len=9)
    tgt = synth.synth_labels(B, Cn, seed=25)
    ctx = sd["token_embedding.weight"][[9, 10, 11, 12]].clone().requires_grad_()
    loss, logits = O.train_step_loss(sd, img, cap, tgt, tl, vl, 0.5, ctx=ctx, text_chunk=3)
    loss.backward()
    out = {"logits": logits.detach().numpy(), "loss": np.array(loss.item()), "dctx": ctx.grad.numpy(),
           "top5": O.jt_topk(logits.detach(), 5).numpy()}
    for i, blk in enumerate(list(tl.values()) + list(vl.values())):
        for p, ab in blk.items():
            for k, t in ab.items():
                out[f"grad.layer_{i}.{p}.{k}"] = t.grad.numpy()
    with torch.no_grad():
        out["img_feat"] = O.encode_image(sd, img, vl, 0.5).numpy()
        out["txt_feat_zs"] = O.encode_text(sd, cap).numpy()
    np.savez_compressed(os.path.join(HERE, "tiny_train_step.npz"), **out)


def block_case():
    """One full-size ViT-B/32 residual block (d=768, L=50, B=2) with the shipped LoRA of vision block 0."""
    g = torch.Generator().manual_seed(31)
    d, L, B, H = 768, 50, 2, 12
    cfg = synth.VIT_B32
    full = synth.synth_state_dict(cfg, seed=1234)
    blk = {k: v.double() for k, v in O._block_params(full, "visual.transformer", 0).items()}
    ck = safe_pkl.load(os.path.join(HERE, "lora_weights.pkl"))
    _, vl = O.split_lora_checkpoint(ck["weights"], "both", "all", "ViT-B/32")
    x = torch.randn(L, B, d, generator=g, dtype=torch.float64).float().double().requires_grad_()  # fp32-exact inputs
    y = O.resblock_forward(x, blk, H, None, vl[0], 0.5)
    dy = torch.randn(L, B, d, generator=g, dtype=torch.float64).float().double()
    y.backward(dy)
    np.savez_compressed(os.path.join(HERE, "vitb32_block0.npz"), x=x.detach().numpy().astype(np.float32),
                        dy=dy.numpy().astype(np.float32), y=y.detach().numpy(), dx=x.grad.numpy())


FULL_B, FULL_C, FULL_P = 8, 16, 0.25
FULL_PROMPT_IDS = [320, 1125, 539, 320]  # "a photo of a" (slow_pace.py:124-131)


def full_case():
    """cfg-2 at FULL DEPTH: ViT-B/32 (12 + 12 blocks, synth seed 1234), the shipped lora_weights.pkl (r=4, q/k/v),
    4 prompt tokens, Philox dropout 0.25 with the engine's first-step seed, B = 8 images / C = 16 captions, one
    run_lora step (lora_train_vlp.py:956-1002) in fp64: loss, logits, top-5 and the flat LoRA + prompt gradient in
    FlatTrainables order (text blocks then vision blocks, per block A_qkv [3r, d] then B_qkv [3d, r]; then ctx)."""
    from clipfs.engine import _mix_seed
    cfg = synth.VIT_B32
    sd = {k: v.double() for k, v in synth.synth_state_dict(cfg, seed=1234).items()}
    ck = safe_pkl.load(os.path.join(HERE, "lora_weights.pkl"))
    tl, vl = O.split_lora_checkpoint(ck["weights"], "both", "all", "ViT-B/32")
    blocks = list(tl.values()) + list(vl.values())
    for blk in blocks:
        for ab in blk.values():
            for t in ab.values():
                t.requires_grad_()
    B, Cn, p = FULL_B, FULL_C, FULL_P
    img = synth.synth_images(B, 224, seed=0).double()
    cap = synth.synth_captions(Cn, 77, cfg.vocab_size, seed=1)
    tgt = synth.synth_labels(B, Cn, seed=2)
    ctx = sd["token_embedding.weight"][FULL_PROMPT_IDS].clone().requires_grad_()
    seed = _mix_seed(0x5EED, 1)  # Engine.seed_base, first step

    def drops(width, seq, n, layers_n, stream0):
        out = {}
        for l in range(layers_n):
            d = {}
            for s, name in enumerate(("q_proj", "k_proj", "v_proj")):
                keep = O.dropout_keep_mask(seed, stream0 + 4 * l + s, n * seq, width, p)
                d[name] = (torch.from_numpy(keep).double() / (1 - p)).reshape(n, seq, width).permute(1, 0, 2)
            out[l] = d
        return out

    td = drops(cfg.transformer_width, 77, Cn, 12, 0)
    vd = drops(cfg.vision_width, 50, B, 12, 1000)
    loss, logits = O.train_step_loss(sd, img, cap, tgt, tl, vl, O.lora_scaling(1, 4), text_drops=td, vis_drops=vd,
                                     ctx=ctx, text_chunk=Cn)
    loss.backward()
    flat = []
    for blk in blocks:
        flat.append(torch.cat([blk[n]["w_lora_A"].grad for n in ("q_proj", "k_proj", "v_proj")], 0).reshape(-1))
        flat.append(torch.cat([blk[n]["w_lora_B"].grad for n in ("q_proj", "k_proj", "v_proj")], 0).reshape(-1))
    flat.append(ctx.grad.reshape(-1))
    flat = torch.cat(flat)
    with torch.no_grad():  # eval-mode (no dropout) logits of the same inputs: the bench's top-5 check uses these
        fi = O.l2_normalize(O.encode_image(sd, img, vl, O.lora_scaling(1, 4)))
        pe = O.build_prompts(ctx.detach(), sd["token_embedding.weight"], cap)
        ft = O.l2_normalize(O.encode_text(sd, cap, tl, O.lora_scaling(1, 4), embeds=pe))
        ev = 100.0 * fi @ ft.t()
        # train-mode (dropout on, same masks) unit features: images and captions are independent units and the masks are
        # indexed by global row, so a B = 256 / C = 403 run must reproduce these in its first 8 / 16 rows
        fi_t = O.l2_normalize(O.encode_image(sd, img, vl, O.lora_scaling(1, 4), drops=vd))
        ft_t = O.l2_normalize(O.encode_text(sd, cap, tl, O.lora_scaling(1, 4), embeds=pe, drops=td))
        assert (100.0 * fi_t @ ft_t.t() - logits.detach()).abs().max().item() < 1e-10
    np.savez_compressed(os.path.join(HERE, "vitb32_full_step.npz"), loss=np.array(loss.item()),
                        logits=logits.detach().numpy(), top5=O.jt_topk(logits.detach().float(), 5).numpy(),
                        flat_grad=flat.numpy().astype(np.float32), grad_max=np.array(flat.abs().max().item()),
                        eval_logits=ev.numpy(), eval_top5=O.jt_topk(ev.float(), 5).numpy(), seed=np.array(seed, dtype=np.uint64),
                        img_feat_train=fi_t.numpy(), txt_feat_train=ft_t.numpy(), img_feat_eval=fi.numpy(),
                        txt_feat_eval=ft.numpy())


L14_B, L14_C, L14_P, L14_R = 4, 8, 0.25, 16
L14_GRAD_STRIDE = 8


def l14_projection_signs(n, seed=77):
    """Seeded +-1 vector: the fixture stores <gradient, signs> per tensor so that EVERY element of the 2.95 M-float
    gradient is covered although only every 8th one is stored."""
    rng = np.random.RandomState(seed)
    return (rng.randint(0, 2, size=n).astype(np.float64) * 2 - 1)


def l14_case():
    """cfg-5 at FULL DEPTH: ViT-L/14 (24 + 12 blocks, synth seed 1234), synthetic rank-16 adapters (seed 5) on the
    reference's placement -- text blocks 0-11, vision blocks 0-20 only (lora_train_vlp.py:57-63) --, 4 prompt tokens,
    Philox dropout 0.25 at the engine's first-step seed, B = 4 images / C = 8 captions, one run_lora step in fp64.
    Stored: loss, logits, top-5, eval-mode (no dropout) logits / top-5 and, of the 2 953 216-float flat LoRA + prompt
    gradient (FlatTrainables order), every 8th element + per-tensor L2 norms + per-tensor signed sums."""
    from clipfs.engine import _mix_seed
    cfg = synth.VIT_L14
    sd = {k: v.double() for k, v in synth.synth_state_dict(cfg, seed=1234).items()}
    lw = synth.synth_lora(cfg, L14_R, seed=5, vision_blocks=range(21))
    tl, vl = O.split_lora_checkpoint(lw, "both", "all", "ViT-L/14")
    assert sorted(vl) == list(range(21)) and sorted(tl) == list(range(12))
    blocks = list(tl.values()) + list(vl.values())
    for blk in blocks:
        for ab in blk.values():
            for t in ab.values():
                t.requires_grad_()
    B, Cn, p = L14_B, L14_C, L14_P
    scaling = O.lora_scaling(1, L14_R)
    img = synth.synth_images(B, 224, seed=0).double()
    cap = synth.synth_captions(Cn, 77, cfg.vocab_size, seed=1)
    tgt = synth.synth_labels(B, Cn, seed=2)
    ctx = sd["token_embedding.weight"][FULL_PROMPT_IDS].clone().requires_grad_()
    seed = _mix_seed(0x5EED, 1)

    def drops(width, seq, n, layer_ids, stream0):
        out = {}
        for l in layer_ids:
            d = {}
            for s, name in enumerate(("q_proj", "k_proj", "v_proj")):
                keep = O.dropout_keep_mask(seed, stream0 + 4 * l + s, n * seq, width, p)
                d[name] = (torch.from_numpy(keep).double() / (1 - p)).reshape(n, seq, width).permute(1, 0, 2)
            out[l] = d
        return out

    td = drops(cfg.transformer_width, 77, Cn, range(12), 0)
    vd = drops(cfg.vision_width, cfg.vision_tokens, B, range(21), 1000)
    loss, logits = O.train_step_loss(sd, img, cap, tgt, tl, vl, scaling, text_drops=td, vis_drops=vd, ctx=ctx,
                                     text_chunk=Cn)
    loss.backward()
    flat, norms, sums = [], [], []
    for blk in blocks:
        for key in ("w_lora_A", "w_lora_B"):
            g = torch.cat([blk[n][key].grad for n in ("q_proj", "k_proj", "v_proj")], 0).reshape(-1)
            flat.append(g)
    flat.append(ctx.grad.reshape(-1))
    off = 0
    for g in flat:
        norms.append(float(g.norm()))
        sums.append(float((g.numpy() * l14_projection_signs(g.numel(), 77 + len(sums))).sum()))
        off += g.numel()
    sizes = np.array([g.numel() for g in flat], dtype=np.int64)
    flat = torch.cat(flat)
    with torch.no_grad():
        fi = O.l2_normalize(O.encode_image(sd, img, vl, scaling))
        pe = O.build_prompts(ctx.detach(), sd["token_embedding.weight"], cap)
        ft = O.l2_normalize(O.encode_text(sd, cap, tl, scaling, embeds=pe))
        ev = 100.0 * fi @ ft.t()
    np.savez_compressed(os.path.join(HERE, "vitl14_full_step.npz"), loss=np.array(loss.item()),
                        logits=logits.detach().numpy(), top5=O.jt_topk(logits.detach().float(), 5).numpy(),
                        flat_grad_strided=flat.numpy()[::L14_GRAD_STRIDE].astype(np.float32),
                        grad_stride=np.array(L14_GRAD_STRIDE), grad_numel=np.array(flat.numel()),
                        grad_max=np.array(flat.abs().max().item()), grad_l2=np.array(float(flat.norm())),
                        tensor_sizes=sizes, tensor_norms=np.array(norms), tensor_signed_sums=np.array(sums),
                        eval_logits=ev.numpy(), eval_top5=O.jt_topk(ev.float(), 5).numpy(),
                        img_feat=fi.numpy(), txt_feat=ft.numpy(), seed=np.array(seed, dtype=np.uint64))


def mta_case():
    g = torch.Generator().manual_seed(41)
    V, d, Cn = 65, 512, 403
    base = torch.randn(1, d, generator=g, dtype=torch.float64)
    feats = O.l2_normalize(base + 0.35 * torch.randn(V, d, generator=g, dtype=torch.float64)).float()
    text = O.l2_normalize(torch.randn(Cn, d, generator=g, dtype=torch.float64) + 0.5 * base).float()
    logits, tr = O.solve_mta(feats, text.t(), return_trace=True)
    mode = O.solve_mta(feats, text.t(), return_mode=True)
    # the same fp32-exact inputs in fp64: what the HIP kernel is held to (north-star: 1e-3 on the logits)
    logits64 = O.solve_mta(feats.double(), text.double().t())
    mode64 = O.solve_mta(feats.double(), text.double().t(), return_mode=True)
    np.savez_compressed(os.path.join(HERE, "mta_v65.npz"), feats=feats.numpy(), text=text.numpy(),
                        logits=logits.numpy(), mode=mode.numpy(), logits64=logits64.numpy(), mode64=mode64.numpy(), y=tr["y"].numpy(), bandwidth=tr["bandwidth"].numpy(),
                        n_y=np.array(tr["n_y"]), n_m=np.array(tr["n_m"]), top5=O.jt_topk(logits, 5).numpy(),
                        is_base=O.ood_is_base(logits).numpy())


def tokenizer_case():
    tk = O.OracleTokenizer(os.path.join(ROOT, "jittor-clip-fewshot_amd", "jclip", "bpe_simple_vocab_16e6.txt.gz"))
    ids = tk.tokenize(SENTENCES).numpy()
    names = [ln.split()[0] for ln in open(os.path.join(HERE, "classes.txt"))]
    prompts = ["a photo of a " + (n.split("_", 1)[1] if "_" in n else n).replace("_", " ") + "." for n in names]
    np.savez_compressed(os.path.join(HERE, "tokenizer_ids.npz"), sentences=np.array(SENTENCES), ids=ids,
                        class_prompt_ids=tk.tokenize(prompts).numpy())


def philox_case():
    keep = O.dropout_keep_mask(0x1234ABCD5, 7, 5, 64, 0.25)
    np.savez_compressed(os.path.join(HERE, "philox_mask.npz"), keep=keep)


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--only-mta" in sys.argv:
        mta_case()
        raise SystemExit(0)
    if "--only-full" in sys.argv:
        full_case()
        raise SystemExit(0)
    if "--only-l14" in sys.argv:
        l14_case()
        raise SystemExit(0)
    tiny_case()
    block_case()
    if "--no-full" not in sys.argv:
        full_case()
        l14_case()
    mta_case()
    tokenizer_case()
    philox_case()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
