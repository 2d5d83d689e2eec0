"""HIP engine vs the COMMITTED golden vectors (tests/golden/*.npz, written by make_golden.py from the fp64
oracle).  Nothing here reads /root/reference or imports the oracle: it is the check that travels."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def test_tiny_train_step_vs_golden(dev, golden_dir):
    import lora_train_vlp as L
    from clipfs import synth
    from jclip.model import build_model
    z = np.load(os.path.join(golden_dir, "tiny_train_step.npz"))
    cfg = synth.TINY
    sd = synth.synth_state_dict(cfg, seed=21, perturb=True)
    model = build_model(sd, device=dev)
    args = types.SimpleNamespace(encoder="both", position="all", backbone="tiny", params=["q", "k", "v", "o"], r=4,
                                 alpha=1, dropout_rate=0.0)
    saved = L.INDEX_POSITIONS_TEXT["all"]
    L.INDEX_POSITIONS_TEXT["all"] = list(range(cfg.transformer_layers))
    L.INDEX_POSITIONS_VISION["tiny"] = {"all": list(range(cfg.vision_layers))}
    try:
        layers = L.apply_lora(args, model)
    finally:
        L.INDEX_POSITIONS_TEXT["all"] = saved
        del L.INDEX_POSITIONS_VISION["tiny"]
    lw = synth.synth_lora(cfg, 4, seed=22, params=("q", "k", "v", "o"))
    names = {"q": "q_proj", "k": "k_proj", "v": "v_proj", "o": "proj"}
    with torch.no_grad():
        for i, layer in enumerate(layers):
            for p in "qkvo":
                m = getattr(layer, names[p])
                m.w_lora_A.copy_(torch.from_numpy(lw[f"layer_{i}"][names[p]]["w_lora_A"]))
                m.w_lora_B.copy_(torch.from_numpy(lw[f"layer_{i}"][names[p]]["w_lora_B"]))
    img = synth.synth_images(5, cfg.image_resolution, seed=23).to(dev)
    cap = synth.synth_captions(7, cfg.context_length, cfg.vocab_size, seed=24, max_len=9).to(dev)
    tgt = synth.synth_labels(5, 7, seed=25).to(dev)
    ctx = torch.nn.Parameter(sd["token_embedding.weight"][[9, 10, 11, 12]].clone().to(dev))
    with torch.no_grad():
        assert np.abs(model.encode_image(img).cpu().numpy() - z["img_feat"]).max() < 2e-5
    tr = L.LoRATrainer(model, prompt_ctx=ctx)
    tr.flat.zero_grad()
    loss_sum, _, logits = tr.forward_backward(img, cap, tgt)
    assert np.abs(logits.cpu().numpy() - z["logits"]).max() < 1e-3   # north-star tolerance on 100 x cosine
    assert abs(loss_sum.item() / 5 - float(z["loss"])) < 1e-4
    assert np.array_equal(L.ops.topk(logits, 5).cpu().numpy(), z["top5"])  # top-5 labels bit-exact
    scale = max(np.abs(z[k]).max() for k in z.files if k.startswith("grad."))
    for i, layer in enumerate(layers):
        slots = {id(p): g for p, g in layer.trainable_pairs()}
        for p in "qkvo":
            m = getattr(layer, names[p])
            for nm, prm in (("w_lora_A", m.w_lora_A), ("w_lora_B", m.w_lora_B)):
                err = np.abs(slots[id(prm)].cpu().numpy() - z[f"grad.layer_{i}.{names[p]}.{nm}"]).max()
                assert err < 1e-4 * scale, (i, p, nm, err, scale)
    assert np.abs(ctx.grad_slot.cpu().numpy() - z["dctx"]).max() < 1e-4 * max(np.abs(z["dctx"]).max(), 1e-3)


def test_vitb32_block_vs_golden(dev, golden_dir):
    """One full-size ViT-B/32 block (d=768, L=50, H=12) with the reference's trained LoRA of vision block 0,
    forward and input-gradient, driven through the C tower ABI."""
    import lora_train_vlp as L
    from clipfs import synth
    from clipfs.engine import _TowerRT
    from jclip.model import Transformer
    z = np.load(os.path.join(golden_dir, "vitb32_block0.npz"))
    full = synth.synth_state_dict(synth.VIT_B32, seed=1234)
    sd = {k: v.to(dev) for k, v in full.items() if k.startswith("visual.transformer.resblocks.0.")}
    tower = Transformer(sd, "visual.transformer", 768, 1, 12, causal=False)
    from clipfs import safe_pkl
    ck = safe_pkl.load(os.path.join(golden_dir, "lora_weights.pkl"))
    mha = L.PlainMultiheadAttentionLoRA(tower.resblocks[0].attn, enable_lora=["q", "k", "v"], r=4, lora_alpha=1,
                                        dropout_rate=0.25)
    tower.resblocks[0].attn = mha
    with torch.no_grad():
        for p in ("q_proj", "k_proj", "v_proj"):
            ab = ck["weights"]["layer_12"][p]  # vision block 0 = layer 12 (text blocks come first)
            getattr(mha, p).w_lora_A.copy_(torch.from_numpy(ab["w_lora_A"]))
            getattr(mha, p).w_lora_B.copy_(torch.from_numpy(ab["w_lora_B"]))
    rt = _TowerRT(tower, 50, stream0=0)
    # golden x is sequence-first [L, B, d]; the engine is token-major [B*L, d]
    x = torch.from_numpy(z["x"]).permute(1, 0, 2).reshape(100, 768).contiguous().to(dev)
    saved = rt.forward(x, 2, True, 0)
    y = x.reshape(2, 50, 768).permute(1, 0, 2).cpu().numpy()
    assert np.abs(y - z["y"]).max() < 2e-5
    dy = torch.from_numpy(z["dy"]).permute(1, 0, 2).reshape(100, 768).contiguous().to(dev)
    rt.backward(dy, 2, saved, 0, stop_at_input=False)
    dx = dy.reshape(2, 50, 768).permute(1, 0, 2).cpu().numpy()
    assert np.abs(dx - z["dx"]).max() < 1e-4


def build_cfg2(dev, golden_dir, dropout):
    """Full ViT-B/32 (synth seed 1234) + the shipped LoRA checkpoint + 4 prompt tokens: the bench's cfg-2 model."""
    import lora_train_vlp as L
    from clipfs import synth
    from jclip.model import build_model
    model = build_model(synth.synth_state_dict(synth.VIT_B32, seed=1234), device=dev)
    args = types.SimpleNamespace(encoder="both", position="all", backbone="ViT-B/32", params=["q", "k", "v"], r=4,
                                 alpha=1, dropout_rate=dropout)
    layers = L.apply_lora(args, model)
    L.load_lora(args, layers, os.path.join(golden_dir, "lora_weights.pkl"))
    L.mark_only_lora_as_trainable(model)
    ctx = torch.nn.Parameter(model.token_embedding.weight.data[torch.tensor([320, 1125, 539, 320], device=dev)].clone())
    return model, ctx


def test_vitb32_full_step_vs_golden(dev, golden_dir):
    """cfg-2 at FULL DEPTH (12 + 12 blocks, shipped LoRA, prompt tokens, Philox dropout 0.25): loss, logits, top-5
    and the whole flat LoRA + prompt gradient of one run_lora step (lora_train_vlp.py:956-1002) against the fp64
    golden -- the 12-block error accumulation of the backward, which the small-model tests cannot show."""
    import lora_train_vlp as L
    from clipfs import synth
    z = np.load(os.path.join(golden_dir, "vitb32_full_step.npz"))
    model, ctx = build_cfg2(dev, golden_dir, 0.25)
    B, Cn = z["logits"].shape
    img = synth.synth_images(B, 224, seed=0).to(dev)
    cap = synth.synth_captions(Cn, 77, synth.VIT_B32.vocab_size, seed=1).to(dev)
    tgt = synth.synth_labels(B, Cn, seed=2).to(dev)
    model.eval()
    tr = L.LoRATrainer(model, prompt_ctx=ctx)
    with torch.no_grad():
        fi = L.ops.l2norm_fwd(model.encode_image(img))
        from clipfs.engine import encode_text
        ft = L.ops.l2norm_fwd(encode_text(model, cap, ctx))
        ev = L.ops.gemm_nt(fi, ft, alpha=100.0)
    assert np.abs(ev.cpu().numpy() - z["eval_logits"]).max() < 1e-3
    assert np.array_equal(L.ops.topk(ev, 5).cpu().numpy(), z["eval_top5"])
    model.train()
    assert model.engine.step == 0  # the golden's dropout seed is the engine's first-step seed
    tr.flat.zero_grad()
    loss_sum, _, logits = tr.forward_backward(img, cap, tgt)
    from clipfs.engine import _mix_seed
    assert _mix_seed(model.engine.seed_base, model.engine.step) == int(z["seed"])
    err = np.abs(logits.cpu().numpy() - z["logits"]).max()
    assert err < 1e-3, err                                            # north-star tolerance on 100 x cosine
    assert abs(loss_sum.item() / B - float(z["loss"])) < 1e-4
    assert np.array_equal(L.ops.topk(logits, 5).cpu().numpy(), z["top5"])  # top-5 labels bit-exact
    g = tr.flat.grads.cpu().numpy()
    assert g.shape == z["flat_grad"].shape
    gerr = np.abs(g - z["flat_grad"]).max()
    assert gerr <= 1e-4 * float(z["grad_max"]), (gerr, float(z["grad_max"]))


def test_cfg2_full_batch_reproduces_the_golden_prefix(dev, golden_dir):
    """cfg-2 AT THE SIZE THE BENCH TIMES: B = 256 images / C = 403 captions, full depth, two tower streams, compact
    last block, Philox dropout 0.25 at the golden's seed (lora_train_vlp.py:940,976: bs 256, every class caption
    re-encoded per step).  Images and captions are independent units and the dropout masks are indexed by GLOBAL row,
    so rows 0..7 / classes 0..15 of this run must reproduce the committed 8 x 16 fp64 fixture: unit features to 1e-5,
    the [8, 16] logit sub-block to 1e-3, its top-5 identical.  This is the launch geometry of the headline number
    (M = 12 800 / 31 031 rows: gemm_nt_kernel<64,128,3> with XCD super-tiles, no split-K), which the 8 x 16 tests
    never reach.  Then the same step with every block dense (`sparse_backward = False`) must agree with it."""
    import lora_train_vlp as L
    from clipfs import synth
    z = np.load(os.path.join(golden_dir, "vitb32_full_step.npz"))
    model, ctx = build_cfg2(dev, golden_dir, 0.25)
    B, Cn = 256, 403
    b0, c0 = z["logits"].shape
    img = synth.synth_images(B, 224, seed=0).to(dev)
    cap = synth.synth_captions(Cn, 77, synth.VIT_B32.vocab_size, seed=1).to(dev)
    tgt = synth.synth_labels(B, 374, seed=2).to(dev)
    model.train()
    tr = L.LoRATrainer(model, prompt_ctx=ctx)
    assert tr.overlap_towers and model.engine.sparse_backward and model.engine.step == 0
    tr.flat.zero_grad()
    loss_sum, _, logits = tr.forward_backward(img, cap, tgt)
    torch.cuda.synchronize()
    from clipfs.engine import _mix_seed
    assert _mix_seed(model.engine.seed_base, model.engine.step) == int(z["seed"])
    img_n, txt = tr.last_features
    assert np.abs(img_n[:b0].cpu().numpy() - z["img_feat_train"]).max() < 1e-5
    assert np.abs(txt[:c0].cpu().numpy() - z["txt_feat_train"]).max() < 1e-5
    sub = logits[:b0, :c0].contiguous()
    err = np.abs(sub.cpu().numpy() - z["logits"]).max()
    assert err < 1e-3, err
    assert np.array_equal(L.ops.topk(sub, 5).cpu().numpy(), z["top5"])
    assert torch.isfinite(logits).all() and torch.isfinite(tr.flat.grads).all()
    g_compact = tr.flat.grads.clone()
    f_compact = (img_n.clone(), txt.clone(), logits.clone(), loss_sum.clone())
    # every block dense in both directions: same features, logits and gradients (only the last block's dead rows differ)
    model.engine.sparse_backward = False
    model.engine.step = 0
    tr.flat.zero_grad()
    loss_d, _, logits_d = tr.forward_backward(img, cap, tgt)
    torch.cuda.synchronize()
    model.engine.sparse_backward = True
    img_d, txt_d = tr.last_features
    assert (img_d - f_compact[0]).abs().max().item() < 2e-6 and (txt_d - f_compact[1]).abs().max().item() < 2e-6
    assert (logits_d - f_compact[2]).abs().max().item() < 2e-4
    assert abs(loss_d.item() - f_compact[3].item()) < 1e-3 * B
    gmax = g_compact.abs().max().item()
    assert (tr.flat.grads - g_compact).abs().max().item() < 1e-4 * gmax
    # the eval-mode (no dropout) logits of the same prefix
    model.eval()
    with torch.no_grad():
        from clipfs.engine import encode_text
        fi = L.ops.l2norm_fwd(model.encode_image(img))
        ft = L.ops.l2norm_fwd(encode_text(model, cap, ctx))
        ev = L.ops.gemm_nt(fi, ft, alpha=100.0)
    assert np.abs(fi[:b0].cpu().numpy() - z["img_feat_eval"]).max() < 1e-5
    assert np.abs(ft[:c0].cpu().numpy() - z["txt_feat_eval"]).max() < 1e-5
    assert np.abs(ev[:b0, :c0].cpu().numpy() - z["eval_logits"]).max() < 1e-3


def build_cfg5(dev, dropout):
    """Full ViT-L/14 (24 + 12 blocks, synth seed 1234) + synthetic rank-16 adapters (seed 5) on the reference's placement
    (text 0-11, vision 0-20: lora_train_vlp.py:57-63) + 4 prompt tokens: the bench's cfg-5 model."""
    import lora_train_vlp as L
    from clipfs import synth
    from jclip.model import build_model
    cfg = synth.VIT_L14
    model = build_model(synth.synth_state_dict(cfg, seed=1234), device=dev)
    args = types.SimpleNamespace(encoder="both", position="all", backbone="ViT-L/14", params=["q", "k", "v"], r=16,
                                 alpha=1, dropout_rate=dropout)
    layers = L.apply_lora(args, model)
    assert len(layers) == 12 + 21
    lw = synth.synth_lora(cfg, 16, seed=5, vision_blocks=range(21))
    with torch.no_grad():
        for i, layer in enumerate(layers):
            for name in ("q_proj", "k_proj", "v_proj"):
                m = getattr(layer, name)
                m.w_lora_A.copy_(torch.from_numpy(lw[f"layer_{i}"][name]["w_lora_A"]))
                m.w_lora_B.copy_(torch.from_numpy(lw[f"layer_{i}"][name]["w_lora_B"]))
    L.mark_only_lora_as_trainable(model)
    ctx = torch.nn.Parameter(model.token_embedding.weight.data[torch.tensor([320, 1125, 539, 320], device=dev)].clone())
    return model, ctx


def l14_signs(n, seed):
    rng = np.random.RandomState(seed)
    return rng.randint(0, 2, size=n).astype(np.float64) * 2 - 1


# fp16 storage mode at FULL depth (24 + 12 blocks): the STATED budget against the fp64 oracle.  Measured on MI355X (round 3,
# printed by the test): logits 4.7e-3 (eval) / 3.3e-3 (train, dropout 0.25), unit features 1.0e-4, top-5 lists 4 / 4 rows,
# gradient 2.5e-3 relative L2 (largest entry off by 4.1e-3 of 2.13), loss 9.6e-4 -- the budget leaves ~4x head-room.
L14_FP16_LOGIT_TOL = 2e-2      # on 100 x cosine logits
L14_FP16_GRAD_REL_L2 = 1e-2    # ||g - g64|| / ||g64|| over the stored every-8th-element sample
L14_FP16_MIN_TOP5_ROWS = 4     # of 4 images: rows whose top-5 label LIST equals the oracle's (gaps >= 0.1 between neighbours)


def test_vitl14_full_depth_vs_golden(dev, golden_dir):
    """cfg-5 at FULL DEPTH: ViT-L/14 24 + 12 blocks, r = 16 adapters on 21 + 12 blocks, prompt tokens, dropout 0.25,
    4 images x 8 captions, one run_lora step against the fp64 golden (tests/golden/vitl14_full_step.npz).  fp32 mode
    must meet the north-star budget (logits 1e-3, top-5 identical, gradient 1e-4 of its largest entry); the fp16
    storage mode (f16 GEMM operands, f16 MFMA attention, fp32 accumulate / residual stream) gets its own STATED budget:
    the 24-block error accumulation that the depth-2 test cannot show."""
    import lora_train_vlp as L
    from clipfs import synth
    from clipfs.engine import _mix_seed, encode_text
    z = np.load(os.path.join(golden_dir, "vitl14_full_step.npz"))
    model, ctx = build_cfg5(dev, 0.25)
    B, Cn = z["logits"].shape
    img = synth.synth_images(B, 224, seed=0).to(dev)
    cap = synth.synth_captions(Cn, 77, synth.VIT_L14.vocab_size, seed=1).to(dev)
    tgt = synth.synth_labels(B, Cn, seed=2).to(dev)
    tr = L.LoRATrainer(model, prompt_ctx=ctx)
    assert tr.flat.numel == int(z["grad_numel"])
    stride = int(z["grad_stride"])
    sizes = z["tensor_sizes"]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    report = {}
    for mode in ("fp32", "fp16"):
        model.engine.precision = mode
        model.eval()
        with torch.no_grad():
            fi = L.ops.l2norm_fwd(model.encode_image(img))
            ft = L.ops.l2norm_fwd(encode_text(model, cap, ctx))
            ev = L.ops.gemm_nt(fi, ft, alpha=100.0)
        model.train()
        model.engine.step = 0  # the golden's masks are the engine's first-step masks
        tr.flat.zero_grad()
        loss_sum, _, logits = tr.forward_backward(img, cap, tgt)
        torch.cuda.synchronize()
        assert _mix_seed(model.engine.seed_base, model.engine.step) == int(z["seed"])
        g = tr.flat.grads.double().cpu().numpy()
        assert np.isfinite(g).all()
        e_eval = float(np.abs(ev.cpu().numpy() - z["eval_logits"]).max())
        e_train = float(np.abs(logits.cpu().numpy() - z["logits"]).max())
        e_feat = max(float(np.abs(fi.cpu().numpy() - z["img_feat"]).max()), float(np.abs(ft.cpu().numpy() - z["txt_feat"]).max()))
        top5 = L.ops.topk(logits, 5).cpu().numpy()
        rows_ok = int(sum(np.array_equal(top5[i], z["top5"][i]) for i in range(B)))
        top1_ok = int((top5[:, 0] == z["top5"][:, 0]).sum())
        gs = g[::stride]
        rel_l2 = float(np.linalg.norm(gs - z["flat_grad_strided"]) / np.linalg.norm(z["flat_grad_strided"]))
        e_max = float(np.abs(gs - z["flat_grad_strided"]).max())
        norm_rel = max(abs(np.linalg.norm(g[offs[i]:offs[i + 1]]) - z["tensor_norms"][i]) / z["tensor_norms"][i]
                       for i in range(len(sizes)))
        sum_err = max(abs(float((g[offs[i]:offs[i + 1]] * l14_signs(int(sizes[i]), 77 + i)).sum()) - z["tensor_signed_sums"][i]) /
                      (z["tensor_norms"][i] * np.sqrt(sizes[i])) for i in range(len(sizes)))
        loss_err = abs(loss_sum.item() / B - float(z["loss"]))
        report[mode] = dict(eval_logit=e_eval, train_logit=e_train, feat=e_feat, top5_rows=rows_ok, top1=top1_ok,
                            grad_rel_l2=rel_l2, grad_max_err=e_max, grad_max=float(z["grad_max"]), tensor_norm_rel=norm_rel,
                            signed_sum_err=sum_err, loss_err=loss_err)
        print(f"[vitl14 full depth] {mode}: {report[mode]}")
    r32, r16 = report["fp32"], report["fp16"]
    assert r32["eval_logit"] < 1e-3 and r32["train_logit"] < 1e-3, r32
    assert r32["top5_rows"] == B, r32
    assert r32["grad_max_err"] <= 1e-4 * r32["grad_max"], r32
    assert r32["tensor_norm_rel"] < 1e-3 and r32["signed_sum_err"] < 1e-4 and r32["loss_err"] < 1e-4, r32
    # the f16 kernels really ran (error above fp32's) and stay inside the stated fp16 budget at full depth
    assert r32["train_logit"] < r16["train_logit"] < L14_FP16_LOGIT_TOL and r16["eval_logit"] < L14_FP16_LOGIT_TOL, r16
    assert r16["top5_rows"] >= L14_FP16_MIN_TOP5_ROWS and r16["top1"] == B, r16
    assert r16["grad_rel_l2"] < L14_FP16_GRAD_REL_L2, r16
    assert r16["loss_err"] < 5e-3, r16


def test_mta_vs_golden(dev, golden_dir):
    from clipfs import ops
    import ood
    z = np.load(os.path.join(golden_dir, "mta_v65.npz"))
    f = torch.from_numpy(z["feats"]).to(dev).unsqueeze(0).repeat(3, 1, 1)
    t = torch.from_numpy(z["text"]).to(dev)
    mode, logits = ops.mta(f, t)
    for i in range(3):
        assert np.abs(mode[i].cpu().numpy() - z["mode64"][0]).max() < 2e-5      # fp64 oracle on the same fp32-exact inputs
        assert np.abs(logits[i].cpu().numpy() - z["logits64"][0]).max() < 1e-3  # north-star tolerance
    assert np.array_equal(ops.topk(logits, 5).cpu().numpy()[0], z["top5"][0])
    pred = ops.topk(logits, 1).long().squeeze(1)
    assert bool((pred <= ood.BASE_BOUNDARY)[0]) == bool(z["is_base"][0])


def test_dropout_mask_vs_golden(dev, golden_dir):
    """Philox stream bit-exactness: recover the keep mask from lora_down on an identity-like probe."""
    from clipfs import ops
    z = np.load(os.path.join(golden_dir, "philox_mask.npz"))
    keep = z["keep"]  # [5, 64], seed 0x1234ABCD5, stream 7, p = 0.25
    rows, width = keep.shape
    got = np.zeros_like(keep)
    x = torch.ones(rows, width, device=dev)
    for c in range(width):
        A = torch.zeros(1, width, device=dev)
        A[0, c] = 1.0
        t = ops.lora_down(x, A, 1, 1, p=0.25, seed=0x1234ABCD5, stream_base=7)
        got[:, c] = (t[:, 0] > 0).cpu().numpy()
    assert np.array_equal(got, keep)
