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


def test_mta_vs_golden(dev, golden_dir):
    from clipfs import ops
    import ood
    z = np.load(os.path.join(golden_dir, "mta_v65.npz"))
    f = torch.from_numpy(z["feats"]).to(dev).unsqueeze(0).repeat(3, 1, 1)
    t = torch.from_numpy(z["text"]).to(dev)
    mode, logits = ops.mta(f, t)
    for i in range(3):
        assert np.abs(mode[i].cpu().numpy() - z["mode"][0]).max() < 2e-5
        assert np.abs(logits[i].cpu().numpy() - z["logits"][0]).max() < 2e-3
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
