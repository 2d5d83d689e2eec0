"""Edge cases of the hot path on the GPU: single-row / single-image batches, ragged tile edges, a caption that fills
all 77 positions (EOT at the last slot), top-k over all classes, MTA at its minimum view count, the reference's
full 513-view MTA, argument errors surfaced as exceptions (never a silent fallback)."""
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


def _err(a, b):
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()


@pytest.mark.parametrize("M,N,K", [(1, 512, 512), (1, 1, 32), (63, 129, 96), (257, 64, 2304), (3, 403, 512), (2, 768, 3072)])
def test_gemm_degenerate_shapes(dev, M, N, K):
    from clipfs import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g, dtype=torch.float64)
    b = torch.randn(N, K, generator=g, dtype=torch.float64)
    bias = torch.randn(N, generator=g, dtype=torch.float64)
    out = ops.gemm_nt(a.float().to(dev), b.float().to(dev), bias=bias.float().to(dev))
    assert _err(out, a @ b.t() + bias) < 2e-4


def test_single_image_single_caption_full_length(dev):
    """B = 1, one caption whose EOT sits in the last of the 77 positions (max length), train step vs oracle."""
    import lora_train_vlp as L
    from clipfs import synth
    from jclip.model import build_model
    from oracle import clip_oracle as O
    cfg = synth.SMALL
    sd = synth.synth_state_dict(cfg, seed=11, perturb=True)
    model = build_model(sd, device=dev)
    args = types.SimpleNamespace(encoder="both", position="all", backbone="small", params=["q", "k", "v"], r=4, alpha=1,
                                 dropout_rate=0.0)
    saved = L.INDEX_POSITIONS_TEXT["all"]
    L.INDEX_POSITIONS_TEXT["all"] = list(range(cfg.transformer_layers))
    L.INDEX_POSITIONS_VISION["small"] = {"all": list(range(cfg.vision_layers))}
    try:
        layers = L.apply_lora(args, model)
    finally:
        L.INDEX_POSITIONS_TEXT["all"] = saved
        del L.INDEX_POSITIONS_VISION["small"]
    lw = synth.synth_lora(cfg, 4, seed=5)
    names = {"q": "q_proj", "k": "k_proj", "v": "v_proj"}
    with torch.no_grad():
        for i, layer in enumerate(layers):
            for p in "qkv":
                m = getattr(layer, names[p])
                m.w_lora_A.copy_(torch.from_numpy(lw[f"layer_{i}"][names[p]]["w_lora_A"]))
                m.w_lora_B.copy_(torch.from_numpy(lw[f"layer_{i}"][names[p]]["w_lora_B"]))
    n = cfg.context_length
    cap = torch.randint(1, cfg.vocab_size - 3, (2, n))
    cap[:, 0] = cfg.vocab_size - 2
    cap[0, n - 1] = cfg.vocab_size - 1          # EOT in the very last slot
    cap[1, 3] = cfg.vocab_size - 1
    cap[1, 4:] = 0                              # shortest useful caption: SOT, 2 tokens, EOT
    img = synth.synth_images(1, cfg.image_resolution, seed=3)
    tgt = torch.tensor([1])
    model.eval()
    tr = L.LoRATrainer(model)
    tr.flat.zero_grad()
    ls, _, logits = tr.forward_backward(img.to(dev), cap.to(dev), tgt.to(dev))
    sd64 = {k: v.double() for k, v in sd.items()}
    nt = cfg.transformer_layers
    conv = lambda d: {p: {k: torch.from_numpy(v).double() for k, v in ab.items()} for p, ab in d.items()}
    tl = {b: conv(lw[f"layer_{b}"]) for b in range(nt)}
    vl = {b: conv(lw[f"layer_{nt + b}"]) for b in range(cfg.vision_layers)}
    loss, wl = O.train_step_loss(sd64, img.double(), cap, tgt, tl, vl, 0.5)
    assert _err(logits, wl) < 1e-3 and abs(ls.item() - loss.item()) < 1e-4
    with pytest.raises(ValueError):
        model.encode_text(cap[:, :10].to(dev))          # wrong context length
    with pytest.raises(ValueError):
        model.encode_image(torch.zeros(1, 3, 32, 32, device=dev))   # wrong resolution


def test_topk_all_classes_and_errors(dev):
    from clipfs import _lib, ops
    z = torch.randn(4, 403, generator=torch.Generator().manual_seed(1))
    lab = ops.topk(z.to(dev), 403).cpu().long()
    assert torch.equal(lab, torch.sort(-z, dim=1, stable=True).indices)
    with pytest.raises(_lib.ClipfsError):
        ops.topk(z.to(dev), 404)
    with pytest.raises(_lib.ClipfsError):
        ops.layernorm_fwd(torch.zeros(4, 6, device=dev), torch.ones(6, device=dev), torch.zeros(6, device=dev))
    with pytest.raises(AssertionError):
        ops.gemm_nt(torch.zeros(4, 8), torch.zeros(4, 8))   # host tensors are refused: no CPU fallback


@pytest.mark.parametrize("V", [5, 513])
def test_mta_view_count_extremes(dev, V):
    """V = 5 is the minimum (k = int(0.3 * 4) = 1); V = 513 is the reference's 1 + 512 views (ood.py:956)."""
    from clipfs import ops
    from oracle import clip_oracle as O
    g = torch.Generator().manual_seed(V)
    d, Cn = 512, 403
    base = torch.randn(1, d, generator=g, dtype=torch.float64)
    feats = O.l2_normalize(base + 0.4 * torch.randn(V, d, generator=g, dtype=torch.float64)).float()
    text = O.l2_normalize(torch.randn(Cn, d, generator=g, dtype=torch.float64) + 0.5 * base).float()
    mode, logits = ops.mta(feats.to(dev).unsqueeze(0), text.to(dev))
    wl = O.solve_mta(feats, text.t())
    wm = O.solve_mta(feats, text.t(), return_mode=True)
    assert _err(mode, wm) < 5e-5 and _err(logits, wl) < 5e-3
    assert torch.equal(ops.topk(logits, 5).cpu().long(), O.jt_topk(wl, 5))
    if V == 5:
        from clipfs import _lib
        with pytest.raises(_lib.ClipfsError):
            ops.mta(feats[:4].to(dev).unsqueeze(0), text.to(dev))
