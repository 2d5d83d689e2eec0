"""Pins the oracle: recomputes every committed golden vector (tests/golden/make_golden.py) and runs the
algebraic self-checks of SURVEY.md section 8c that need no second implementation."""
import os

import numpy as np
import pytest
import torch

from clipfs import safe_pkl, synth
from oracle import clip_oracle as O


def test_philox_known_answer_vectors():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = O.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(x) for x in got) == want


def test_philox_mask_fixture_and_rate(golden_dir):
    z = np.load(os.path.join(golden_dir, "philox_mask.npz"))
    assert np.array_equal(O.dropout_keep_mask(0x1234ABCD5, 7, 5, 64, 0.25), z["keep"])
    keep = O.dropout_keep_mask(99, 3, 512, 512, 0.25)
    assert abs(keep.mean() - 0.75) < 5e-3
    assert not np.array_equal(O.dropout_keep_mask(99, 4, 8, 64, 0.25), O.dropout_keep_mask(99, 3, 8, 64, 0.25))


def test_tiny_train_step_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "tiny_train_step.npz"))
    cfg = synth.TINY
    sd = {k: v.double() for k, v in synth.synth_state_dict(cfg, seed=21, perturb=True).items()}
    lw = synth.synth_lora(cfg, 4, seed=22, params=("q", "k", "v", "o"))
    nt = cfg.transformer_layers
    conv = lambda d: {p: {k: torch.from_numpy(v).double().requires_grad_() for k, v in ab.items()} for p, ab in d.items()}
    tl = {b: conv(lw[f"layer_{b}"]) for b in range(nt)}
    vl = {b: conv(lw[f"layer_{nt + b}"]) for b in range(cfg.vision_layers)}
    img = synth.synth_images(5, cfg.image_resolution, seed=23).double()
    cap = synth.synth_captions(7, cfg.context_length, cfg.vocab_size, seed=24, max_len=9)
    tgt = synth.synth_labels(5, 7, seed=25)
    ctx = sd["token_embedding.weight"][[9, 10, 11, 12]].clone().requires_grad_()
    # chunking the captions (encode_text_in_batches, lora_train_vlp.py:907-919) must not change anything
    loss, logits = O.train_step_loss(sd, img, cap, tgt, tl, vl, 0.5, ctx=ctx, text_chunk=32)
    loss.backward()
    assert np.allclose(logits.detach().numpy(), z["logits"], atol=1e-10)
    assert abs(loss.item() - float(z["loss"])) < 1e-12
    assert np.allclose(ctx.grad.numpy(), z["dctx"], atol=1e-12)
    assert np.array_equal(O.jt_topk(logits.detach(), 5).numpy(), z["top5"])
    assert np.allclose(tl[0]["q_proj"]["w_lora_A"].grad.numpy(), z["grad.layer_0.q_proj.w_lora_A"], atol=1e-12)
    assert np.allclose(vl[1]["proj"]["w_lora_B"].grad.numpy(), z[f"grad.layer_{nt + 1}.proj.w_lora_B"], atol=1e-12)


def test_block_fixture_and_lora_forms(golden_dir):
    """Full-size ViT-B/32 block with the shipped LoRA of vision block 0; merged-weight LoRA (W + s BA,
    lora_train_vlp.py:287-294) == additive LoRA (:296-306) at p = 0; packed QKV == split q/k/v."""
    z = np.load(os.path.join(golden_dir, "vitb32_block0.npz"))
    full = synth.synth_state_dict(synth.VIT_B32, seed=1234)
    blk = {k: v.double() for k, v in O._block_params(full, "visual.transformer", 0).items()}
    ck = safe_pkl.load(os.path.join(golden_dir, "lora_weights.pkl"))
    _, vl = O.split_lora_checkpoint(ck["weights"], "both", "all", "ViT-B/32")
    x = torch.from_numpy(z["x"]).double()
    y = O.resblock_forward(x, blk, 12, None, vl[0], 0.5)
    assert np.allclose(y.numpy(), z["y"], atol=1e-10)
    merged = dict(blk)
    w = blk["in_proj_weight"].clone()
    for i, p in enumerate(("q_proj", "k_proj", "v_proj")):
        ab = vl[0][p]
        w[i * 768:(i + 1) * 768] += 0.5 * ab["w_lora_B"] @ ab["w_lora_A"]
    merged["in_proj_weight"] = w
    y2 = O.resblock_forward(x, merged, 12, None)  # un-adapted path = packed in-projection (mha.py:129-146)
    assert np.allclose(y2.numpy(), z["y"], atol=1e-9)
    a = vl[0]["q_proj"]
    xm = x.reshape(-1, 768)
    l1 = O.lora_linear(xm, blk["in_proj_weight"][:768], blk["in_proj_bias"][:768], a["w_lora_A"], a["w_lora_B"], 0.5)
    l2 = O.lora_linear_merged(xm, blk["in_proj_weight"][:768], blk["in_proj_bias"][:768], a["w_lora_A"], a["w_lora_B"], 0.5)
    assert torch.allclose(l1, l2, atol=1e-10)


def test_mta_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "mta_v65.npz"))
    f, t = torch.from_numpy(z["feats"]), torch.from_numpy(z["text"])
    logits, tr = O.solve_mta(f, t.t(), return_trace=True)
    assert np.allclose(logits.numpy(), z["logits"], atol=1e-4)
    assert np.array_equal(O.jt_topk(logits, 5).numpy(), z["top5"])
    assert np.allclose(O.solve_mta(f, t.t(), return_mode=True).numpy(), z["mode"], atol=1e-6)
    assert tr["n_y"] == list(z["n_y"]) and tr["n_m"] == list(z["n_m"])
    # invariants: y is a distribution over views, mode is a unit vector, bandwidth positive
    assert abs(tr["y"].sum().item() - 1) < 1e-5 and abs(tr["mode"].norm().item() - 1) < 1e-5
    assert (tr["bandwidth"] > 0).all()
    assert bool(O.ood_is_base(logits)[0]) == bool(z["is_base"][0])


def test_causal_text_features_ignore_tokens_after_eot():
    cfg = synth.TINY
    sd = {k: v.double() for k, v in synth.synth_state_dict(cfg, seed=3, perturb=True).items()}
    cap = synth.synth_captions(4, cfg.context_length, cfg.vocab_size, seed=5, max_len=6)
    cap2 = cap.clone()
    eot = cap.argmax(dim=-1)
    for i in range(4):
        cap2[i, eot[i] + 1:] = 7  # garbage after EOT (smaller than the EOT id)
    assert torch.allclose(O.encode_text(sd, cap), O.encode_text(sd, cap2), atol=1e-12)


def test_ood_boundary_and_cls_acc():
    out = torch.full((4, 403), -1.0)
    out[0, 372] = 1
    out[1, 373] = 1
    out[2, 0] = 1
    out[3, 402] = 1
    assert O.ood_is_base(out).tolist() == [True, False, True, False]
    tgt = torch.tensor([10, 400, 380, 402])
    assert O.cls_acc_ood(out, tgt) == 75.0
    assert O.cls_acc(out, torch.tensor([372, 373, 1, 402])) == 75.0


def test_logit_normalize_and_std_clamp():
    z = torch.randn(6, 9, dtype=torch.float64)
    n = O.logit_normalize(z)
    assert torch.allclose(n.mean(dim=1), torch.zeros(6, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(O.jt_std(z), z.flatten().std(unbiased=True))
    assert O.jt_std(torch.ones(3, 3)).item() == pytest.approx(1e-3)  # sqrt(clamp(0, 1e-6))


def test_full_depth_fixture_eval_logits(golden_dir):
    """vitb32_full_step.npz: the dropout-free logits / top-5 of the full-depth cfg-2 model are recomputed here (the
    train-step part of the fixture takes the oracle ~20 s in fp64 and is regenerated by make_golden.py only)."""
    z = np.load(os.path.join(golden_dir, "vitb32_full_step.npz"))
    cfg = synth.VIT_B32
    sd = {k: v.double() for k, v in synth.synth_state_dict(cfg, seed=1234).items()}
    ck = safe_pkl.load(os.path.join(golden_dir, "lora_weights.pkl"))
    tl, vl = O.split_lora_checkpoint(ck["weights"], "both", "all", "ViT-B/32")
    B, Cn = z["eval_logits"].shape
    img = synth.synth_images(B, 224, seed=0).double()
    cap = synth.synth_captions(Cn, 77, cfg.vocab_size, seed=1)
    ctx = sd["token_embedding.weight"][[320, 1125, 539, 320]]
    with torch.no_grad():
        fi = O.l2_normalize(O.encode_image(sd, img, vl, 0.5))
        ft = O.l2_normalize(O.encode_text(sd, cap, tl, 0.5, embeds=O.build_prompts(ctx, sd["token_embedding.weight"], cap)))
        ev = 100.0 * fi @ ft.t()
    assert np.allclose(ev.numpy(), z["eval_logits"], atol=1e-9)
    assert np.array_equal(O.jt_topk(ev.float(), 5).numpy(), z["eval_top5"])
    # 12 text + 12 vision blocks x (A_qkv [3r, d] + B_qkv [3d, r]) at r = 4, plus the 4 x 512 prompt tokens
    assert z["flat_grad"].size == 12 * 2 * 3 * 4 * 512 + 12 * 2 * 3 * 4 * 768 + 4 * 512


def test_vitl14_full_depth_fixture(golden_dir):
    """vitl14_full_step.npz (cfg-5 at full depth): one image and two captions of its dropout-free forward are recomputed
    here in fp64 (the train step with the 2.95 M-float gradient is regenerated by make_golden.py only); the gradient
    bookkeeping (sizes, stride sample, per-tensor norms) must be self-consistent."""
    z = np.load(os.path.join(golden_dir, "vitl14_full_step.npz"))
    cfg = synth.VIT_L14
    sd = {k: v.double() for k, v in synth.synth_state_dict(cfg, seed=1234).items()}
    lw = synth.synth_lora(cfg, 16, seed=5, vision_blocks=range(21))
    tl, vl = O.split_lora_checkpoint(lw, "both", "all", "ViT-L/14")
    assert sorted(vl) == list(range(21))  # the reference adapts vision blocks 0-20 only (lora_train_vlp.py:62)
    img = synth.synth_images(4, 224, seed=0)[:1].double()
    cap = synth.synth_captions(8, 77, cfg.vocab_size, seed=1)[:2]
    ctx = sd["token_embedding.weight"][[320, 1125, 539, 320]]
    s = O.lora_scaling(1, 16)
    assert s == 0.25
    with torch.no_grad():
        fi = O.l2_normalize(O.encode_image(sd, img, vl, s))
        ft = O.l2_normalize(O.encode_text(sd, cap, tl, s, embeds=O.build_prompts(ctx, sd["token_embedding.weight"], cap)))
    assert np.allclose(fi.numpy(), z["img_feat"][:1], atol=1e-10)
    assert np.allclose(ft.numpy(), z["txt_feat"][:2], atol=1e-10)
    assert np.allclose((100.0 * fi @ ft.t()).numpy(), z["eval_logits"][:1, :2], atol=1e-8)
    sizes = z["tensor_sizes"]
    assert len(sizes) == 2 * (12 + 21) + 1 and int(sizes.sum()) == int(z["grad_numel"]) == 12 * 2 * 48 * 768 + 21 * 2 * 48 * 1024 + 4 * 768
    assert z["flat_grad_strided"].size == -(-int(z["grad_numel"]) // int(z["grad_stride"]))
    assert abs(float(np.sqrt((z["tensor_norms"] ** 2).sum())) - float(z["grad_l2"])) < 1e-9 * float(z["grad_l2"])
