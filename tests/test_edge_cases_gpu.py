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


def _attn_ref(qkv, B, L, H, causal):
    x = qkv.double().view(B, L, 3, H, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.full((L, L), float("-inf"), dtype=torch.float64).triu(1)
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, H * 64)


@pytest.mark.parametrize("B,L,H,causal", [(1, 1, 1, False), (1, 1, 1, True), (3, 31, 1, True), (1, 32, 3, False),
                                          (2, 33, 1, True), (1, 64, 1, True), (1, 65, 2, False), (5, 2, 1, True)])
def test_mfma_attention_tile_edges(dev, B, L, H, causal):
    """fp32 MFMA attention at the 32-token tile boundaries (one token, one short tile, exactly one / two tiles, one
    token into the next tile), odd batch*heads; forward, log-sum-exp and backward against fp64 autograd."""
    from clipfs import ops
    g = torch.Generator().manual_seed(L * 3 + H)
    qkv = torch.randn(B * L, 3 * H * 64, generator=g, dtype=torch.float64).requires_grad_()
    do = torch.randn(B * L, H * 64, generator=g, dtype=torch.float64)
    ref = _attn_ref(qkv, B, L, H, causal)
    (ref * do).sum().backward()
    qd = qkv.detach().float().to(dev)
    out, lse = ops.attention_fwd(qd, B, L, H, causal, want_lse=True)
    assert _err(out, ref) < 2e-5
    dq = ops.attention_bwd(qd, do.float().to(dev), B, L, H, causal, out=out, lse=lse)
    assert _err(dq, qkv.grad) < 5e-5 * max(1.0, qkv.grad.abs().max().item())


@pytest.mark.parametrize("rows", [1, 15, 16, 17, 65])
def test_mfma_lora_row_edges(dev, rows):
    """fp32-MFMA LoRA kernels with row counts around the 16-row MFMA tile and the 64-row reduction slice."""
    from clipfs import ops
    width, r, nseg = 128, 16, 3
    g = torch.Generator().manual_seed(rows)
    R = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    x, A, Bm, dy = R(rows, width), R(nseg * r, width) * 0.1, R(nseg * width, r) * 0.1, R(rows, nseg * width)
    xs, As, Bs = x.clone().requires_grad_(), A.clone().requires_grad_(), Bm.clone().requires_grad_()
    t = torch.cat([xs @ As[s * r:(s + 1) * r].t() for s in range(nseg)], 1)
    y = torch.cat([0.25 * t[:, s * r:(s + 1) * r] @ Bs[s * width:(s + 1) * width].t() for s in range(nseg)], 1)
    y.backward(dy)
    D = lambda v: v.detach().float().to(dev)
    tg = ops.lora_down(D(x), D(A), r, nseg)
    assert _err(tg, t) < 2e-5
    dA, dB, dx = torch.zeros(nseg * r, width, device=dev), torch.zeros(nseg * width, r, device=dev), torch.zeros(rows, width, device=dev)
    ops.lora_bwd(D(dy), D(x), tg, D(A), D(Bm), dA, dB, dx=dx, scale=0.25)
    tol = lambda w: 2e-5 * max(1.0, w.abs().max().item())
    assert _err(dA, As.grad) < tol(As.grad) and _err(dB, Bs.grad) < tol(Bs.grad) and _err(dx, xs.grad) < tol(xs.grad)


@pytest.mark.parametrize("M,N,K", [(1, 128, 32), (255, 8, 64), (256, 136, 96), (513, 128, 1024), (33024, 128, 32)])
def test_gemm_f16_ragged_shapes(dev, M, N, K):
    """f16 x f16 kernel at ragged tile edges (single row, N below one tile, one row past a 256-row block, the peeled
    tail launch of a long M) with every output kind."""
    from clipfs import ops
    g = torch.Generator().manual_seed(M + N)
    a16 = torch.randn(M, K, generator=g).half().to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    w16 = ops.to_f16(w)
    bias = torch.randn(N, generator=g).to(dev)
    ref = a16.float() @ w16.float().t() + bias
    out16 = torch.empty(M, N, device=dev, dtype=torch.float16)
    out = ops.gemm_nt(None, w, bias=bias, b_planes=w16, a16=a16, out16=out16)
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() <= 2e-5 * scale + 1e-5
    assert (out16.float() - ref).abs().max().item() <= 1e-3 * scale + 1e-4
    only = torch.empty(M, N, device=dev, dtype=torch.float16)
    ops.gemm_nt(None, w, None, bias=bias, b_planes=w16, a16=a16, out16=only, only16=True)
    assert torch.equal(only, out16)
