"""Kernel-level parity: every HIP kernel (through the C ABI) against the CPU oracle / an fp64 torch
restatement of the same op on the same seeded inputs.  Tolerances are absolute on O(1) data and are
stated per test; GEMM-family kernels are exact-f32 MFMA fma chains, so errors are ~1e-6 * sqrt(K)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


def _close(got, want, atol, what=""):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    err = (got - want).abs().max().item()
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol:.1e}"
    return err


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(64, 128, 32), (200, 768, 768), (77 * 5, 192, 64), (130, 403, 512), (50, 64, 100),
                                   (1000, 2304, 768)])
def test_gemm_plain(dev, M, N, K):
    from clipfs import ops
    a, b = _rand(M, K, seed=1), _rand(N, K, seed=2)
    out = ops.gemm_nt(a.float().to(dev), b.float().to(dev))
    _close(out, a @ b.t(), 2e-5 * math.sqrt(K) * 4, f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(12800, 768, 768), (12800, 2304, 256), (12801, 768, 512), (31031, 512, 128)])
def test_gemm_full_size_rows(dev, M, N, K):
    """The stand-alone GEMM at the row counts the bench times (12 800 image-token rows, 31 031 caption-token rows, one
    ragged): several rounds of 64 x 128 tiles over the CUs in the XCD-contiguous super-tile order, against an fp64 product,
    with every epilogue option, bitwise reproducible."""
    from clipfs import ops
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g) * K ** -0.5
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ad, bd = a.to(dev), b.to(dev)
    want = a.double() @ b.double().t()
    out = ops.gemm_nt(ad, bd)
    _close(out, want, 2e-5 * math.sqrt(K) * 4, "full-size plain")
    assert torch.equal(ops.gemm_nt(ad, bd), out)
    u = torch.empty(M, N, device=dev)
    out = ops.gemm_nt(ad, bd, bias=bias.to(dev), residual=res.to(dev), act=1, aux_out=u)
    pre = want + bias.double()
    _close(u, pre, 1e-4, "full-size pre-activation")
    _close(out, pre * torch.sigmoid(1.702 * pre) + res.double(), 1e-4, "full-size bias + gelu + residual")


def test_gemm_asymmetric_layout(dev):
    """A = I with an asymmetric B catches a transposed C write (cdna guide section 3)."""
    from clipfs import ops
    n = 128
    a = torch.eye(n, dtype=torch.float32)
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 1000.0
    out = ops.gemm_nt(a.to(dev), b.to(dev))
    assert torch.equal(out.cpu(), b.t().contiguous())


def test_gemm_epilogues(dev):
    from clipfs import ops
    from oracle import clip_oracle as O
    M, N, K, r = 300, 384, 128, 4
    a, w = _rand(M, K, seed=3), _rand(N, K, seed=4, scale=K ** -0.5)
    bias, res = _rand(N, seed=5), _rand(M, N, seed=6)
    t, lb = _rand(M, 3 * r, seed=7), _rand(N, r, seed=8)
    D = lambda x: x.float().to(dev)
    # bias + LoRA up-projection + residual
    out = ops.gemm_nt(D(a), D(w), bias=D(bias), residual=D(res), lora_t=D(t), lora_b=D(lb), lora_seg_width=128,
                      lora_scale=0.5)
    want = a @ w.t() + bias + res
    for s in range(3):
        want[:, s * 128:(s + 1) * 128] += 0.5 * t[:, s * r:(s + 1) * r] @ lb[s * 128:(s + 1) * 128].t()
    _close(out, want, 1e-4, "gemm bias+lora+residual")
    # QuickGELU with saved pre-activation
    u = torch.empty(M, N, device=dev)
    out = ops.gemm_nt(D(a), D(w), bias=D(bias), act=1, aux_out=u)
    pre = a @ w.t() + bias
    _close(u, pre, 1e-4, "gemm pre-activation")
    _close(out, O.quick_gelu(pre), 1e-4, "gemm quickgelu")
    # backward of QuickGELU fused on a dgrad GEMM
    up = _rand(M, N, seed=9).requires_grad_()
    (O.quick_gelu(up)).sum().backward()
    out = ops.gemm_nt(D(a), D(w), act=2, aux_in=D(up.detach()))
    _close(out, (a @ w.t()) * up.grad, 1e-4, "gemm gelu-grad")
    # alpha
    out = ops.gemm_nt(D(a), D(w), alpha=100.0)
    _close(out, 100.0 * (a @ w.t()), 2e-3, "gemm alpha")


@pytest.mark.parametrize("B,R,ps,width", [(3, 64, 32, 128), (2, 224, 32, 768), (2, 28, 14, 64)])
def test_patch_embed(dev, B, R, ps, width):
    from clipfs import ops
    img = _rand(B, 3, R, R, seed=1)
    w = _rand(width, 3, ps, ps, seed=2, scale=(3 * ps * ps) ** -0.5)
    P = (R // ps) ** 2
    L = P + 1
    pos = _rand(L, width, seed=3)
    x = torch.zeros(B * L, width, device=dev)
    ops.patch_embed(img.float().to(dev), w.float().to(dev), pos.float().to(dev), x, L)
    conv = torch.nn.functional.conv2d(img, w, stride=ps).reshape(B, width, P).permute(0, 2, 1)
    want = torch.zeros(B, L, width, dtype=torch.float64)
    want[:, 1:] = conv + pos[1:]
    _close(x.reshape(B, L, width), want, 1e-4, "patch embed")


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,width", [(7, 64), (1001, 768), (403, 512), (5, 1024)])
def test_layernorm(dev, rows, width):
    from clipfs import ops
    from oracle import clip_oracle as O
    x = (_rand(rows, width, seed=1) * 3 + 1).requires_grad_()
    g, b = 1 + 0.1 * _rand(width, seed=2), _rand(width, seed=3)
    D = lambda t: t.detach().float().to(dev)
    y, mean, rstd = ops.layernorm_fwd(D(x), D(g), D(b), save_stats=True)
    want = O.jt_layer_norm(x, g, b)
    _close(y, want, 2e-5, "ln fwd")
    dy, dres = _rand(rows, width, seed=4), _rand(rows, width, seed=5)
    want.backward(dy)
    dx = ops.layernorm_bwd(D(dy), D(x), D(g), mean, rstd, dres=D(dres))
    _close(dx, x.grad + dres, 5e-5, "ln bwd")


def test_layernorm_strided_rows(dev):
    """ln_post reads the class-token rows of [B, L, d] in place (ldx = L*d)."""
    from clipfs import ops
    from oracle import clip_oracle as O
    B, L, d = 6, 5, 128
    x = _rand(B, L, d, seed=1)
    g, b = 1 + 0.1 * _rand(d, seed=2), _rand(d, seed=3)
    D = lambda t: t.float().to(dev)
    y = ops.layernorm_fwd(D(x), D(g), D(b), ldx=L * d, rows=B)
    _close(y, O.jt_layer_norm(x[:, 0], g, b), 2e-5, "ln strided")


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,L,H,causal", [(3, 50, 2, False), (2, 54, 12, False), (3, 77, 8, True), (2, 16, 1, True),
                                          (1, 5, 2, False), (2, 96, 2, True), (2, 257, 2, False), (1, 130, 1, True),
                                          (1, 100, 2, False), (1, 288, 1, True), (1, 330, 2, False), (1, 289, 1, True)])
def test_attention(dev, B, L, H, causal):
    from clipfs import ops
    from oracle import clip_oracle as O
    d = H * 64
    qkv = _rand(B * L, 3 * d, seed=1).requires_grad_()
    q, k, v = [qkv[:, i * d:(i + 1) * d].reshape(B, L, H, 64).permute(0, 2, 1, 3) for i in range(3)]
    mask = O.build_causal_mask(L, torch.float64) if causal else None
    o = O.sdpa(q, k, v, mask).permute(0, 2, 1, 3).reshape(B * L, d)
    qd = qkv.detach().float().to(dev)
    got, lse = ops.attention_fwd(qd, B, L, H, causal, want_lse=True)  # fp32 MFMA kernels for L <= 288, streaming above
    assert lse is not None
    ref_lse = torch.logsumexp((q @ k.transpose(-1, -2)) * 0.125 + (mask if causal else 0), -1).reshape(-1)
    _close(lse, ref_lse.detach(), 2e-5, "attention lse")
    _close(got, o, 2e-5, "attention fwd")
    _close(ops.attention_fwd(qd, B, L, H, causal), o, 2e-5, "attention fwd (inference)")
    do = _rand(B * L, d, seed=2)
    o.backward(do)
    dq = ops.attention_bwd(qd, do.float().to(dev), B, L, H, causal, out=got, lse=lse)
    _close(dq, qkv.grad, 5e-5, "attention bwd")
    if L <= 96:  # without the forward's lse the backward falls back to the softmax-recomputing VALU kernels
        dq2 = ops.attention_bwd(qd, do.float().to(dev), B, L, H, causal)
        _close(dq2, qkv.grad, 5e-5, "attention bwd (no lse)")


# ------------------------------------------------------------------ LoRA
@pytest.mark.parametrize("p", [0.0, 0.25])
@pytest.mark.parametrize("rows,width,r,nseg", [(333, 128, 4, 3), (200, 256, 16, 3), (77, 128, 16, 1), (150, 192, 2, 3),
                                                (1100, 128, 8, 3), (90, 384, 1, 1), (257, 512, 4, 3), (130, 768, 12, 3),
                                                (45, 1024, 16, 3), (61, 256, 3, 1)])
def test_lora_down_and_bwd(dev, p, rows, width, r, nseg):
    """width % 128 == 0 runs the fp32-MFMA kernels (lora_mfma.hip), 192 the one-wave-per-row kernels (lora.hip); the
    512 / 768 / 1024 cases are the towers' own widths."""
    from clipfs import ops
    from oracle import clip_oracle as O
    seed, sb = 0x1234ABCD5, 7
    x = _rand(rows, width, seed=1)
    A = _rand(nseg * r, width, seed=2, scale=width ** -0.5).requires_grad_()
    Bm = _rand(nseg * width, r, seed=3, scale=0.1).requires_grad_()
    xs = x.clone().requires_grad_()
    scale = 0.5
    y = torch.zeros(rows, nseg * width, dtype=torch.float64)
    ts = []
    parts = []
    for s in range(nseg):
        m = torch.ones(rows, width, dtype=torch.float64)
        if p > 0:
            keep = O.dropout_keep_mask(seed, sb + s, rows, width, p)
            m = torch.from_numpy(keep).double() / (1 - p)
        t = (xs * m) @ A[s * r:(s + 1) * r].t()
        ts.append(t)
        parts.append(scale * t @ Bm[s * width:(s + 1) * width].t())
    y = torch.cat(parts, dim=1)
    D = lambda t: t.detach().float().to(dev)
    t_gpu = ops.lora_down(D(x), D(A), r, nseg, p=p, seed=seed if p > 0 else 0, stream_base=sb)
    _close(t_gpu, torch.cat(ts, dim=1), 2e-5, "lora down")
    dy = _rand(rows, nseg * width, seed=4)
    y.backward(dy)
    dA = torch.zeros(nseg * r, width, device=dev)
    dB = torch.zeros(nseg * width, r, device=dev)
    dx = torch.zeros(rows, width, device=dev)
    ops.lora_bwd(D(dy), D(x), t_gpu, D(A), D(Bm), dA, dB, dx=dx, scale=scale, p=p, seed=seed if p > 0 else 0,
                 stream_base=sb)
    _close(dA, A.grad, 2e-4, "lora dA")
    _close(dB, Bm.grad, 2e-4, "lora dB")
    _close(dx, xs.grad, 2e-4, "lora dx")
    # the forward's masks recorded as keep bits and read back by the backward (matrix-core shapes): the recorded bits are
    # the oracle's masks, and the backward that reads them gives the bits of the backward that regenerates them
    from clipfs import _lib
    if p > 0 and _lib.load().clipfs_lora_keep_bits_ok(width, width, r, nseg):
        kb = ops.lora_keep_bits(rows, width, dev)
        t2 = ops.lora_down(D(x), D(A), r, nseg, p=p, seed=seed, stream_base=sb, keep_bits=kb)
        assert torch.equal(t2, t_gpu)
        bits = kb.cpu().numpy().view(np.uint16)
        for s in range(nseg):
            keep = O.dropout_keep_mask(seed, sb + s, rows, width, p).reshape(rows, width // 4, 4)
            got = np.stack([(bits >> (4 * s + e)) & 1 for e in range(4)], axis=-1).astype(bool)
            assert np.array_equal(got, keep), f"keep bits of segment {s}"
        dA2, dB2, dx2 = torch.zeros_like(dA), torch.zeros_like(dB), torch.zeros_like(dx)
        ops.lora_bwd(D(dy), D(x), t_gpu, D(A), D(Bm), dA2, dB2, dx=dx2, scale=scale, p=p, seed=seed, stream_base=sb, keep_bits=kb)
        assert torch.equal(dA2, dA) and torch.equal(dB2, dB) and torch.equal(dx2, dx)


@pytest.mark.gpu
@pytest.mark.parametrize("p", [0.0, 0.25])
@pytest.mark.parametrize("rows,width,r,mask", [(333, 768, 4, 7), (1601, 512, 4, 5), (50, 1024, 2, 7), (7, 192, 1, 2)])
def test_layernorm_with_lora_down(dev, p, rows, width, r, mask):
    """LayerNorm + adapter down-projection in one pass (clipfs_layernorm_fwd_lora, the small-rank path of the tower) against
    an fp64 restatement with the oracle's Philox masks: y and the statistics are the plain LayerNorm's bits, t is the
    down-projection of dropout(y) to fp32 rounding (2e-5 of the largest entry); segments outside the mask are zero."""
    from clipfs import ops
    from oracle import clip_oracle as O
    seed, sb, row0 = 0x5EED1234, 11, 40
    x = _rand(rows, width, seed=5, scale=2.0) + 0.3
    g = _rand(width, seed=6) * 0.1 + 1.0
    b = _rand(width, seed=7) * 0.1
    A = _rand(3 * r, width, seed=8, scale=width ** -0.5)
    D = lambda t: t.detach().float().to(dev)
    y, t, mean, rstd = ops.layernorm_fwd_lora(D(x), D(g), D(b), D(A), r, 3, seg_mask=mask, p=p, seed=seed if p > 0 else 0,
                                              stream_base=sb, row0=row0)
    y0, m0, r0 = ops.layernorm_fwd(D(x), D(g), D(b), save_stats=True)
    assert torch.equal(y, y0) and torch.equal(mean, m0) and torch.equal(rstd, r0)
    yd = y.double().cpu()
    ref = torch.zeros(rows, 3 * r, dtype=torch.float64)
    for s in range(3):
        if not (mask >> s) & 1:
            continue
        m = torch.ones(rows, width, dtype=torch.float64)
        if p > 0:
            keep = O.dropout_keep_mask(seed, sb + s, rows + row0, width, p)[row0:]
            m = torch.from_numpy(keep).double() / (1 - p)
        ref[:, s * r:(s + 1) * r] = (yd * m) @ A[s * r:(s + 1) * r].double().t()
    _close(t, ref, 2e-5, "fused lora down")
    for s in range(3):
        if not (mask >> s) & 1:
            assert t[:, s * r:(s + 1) * r].abs().max().item() == 0.0
    # and the stand-alone kernel on the same y (what the backward's masks were tested against)
    t1 = ops.lora_down(y, D(A), r, 3, seg_mask=mask, p=p, seed=seed if p > 0 else 0, stream_base=sb, row0=row0)
    _close(t, t1.double().cpu(), 2e-5, "fused vs stand-alone down")
    if p > 0 and width % 128 == 0:  # both forwards record the same keep bits
        kb0, kb1 = ops.lora_keep_bits(rows, width, dev), ops.lora_keep_bits(rows, width, dev)
        ops.layernorm_fwd_lora(D(x), D(g), D(b), D(A), r, 3, seg_mask=mask, p=p, seed=seed, stream_base=sb, row0=row0, keep_bits=kb0)
        ops.lora_down(y, D(A), r, 3, seg_mask=mask, p=p, seed=seed, stream_base=sb, row0=row0, keep_bits=kb1)
        assert torch.equal(kb0, kb1) and kb0.any()


def test_dropout_rate(dev):
    """Philox keep-rate of the device stream is 1 - p (statistical sanity, 1e6 draws)."""
    from clipfs import ops
    rows, width = 2048, 512
    x = torch.ones(rows, width, device=dev)
    A = torch.ones(1, width, device=dev)
    t = ops.lora_down(x, A, 1, 1, p=0.25, seed=99, stream_base=3)  # = kept_count / 0.75 per row
    kept = (t.double().sum() * 0.75).item() / (rows * width)
    assert abs(kept - 0.75) < 2e-3


# ------------------------------------------------------------------ token assembly / head / loss
def test_text_embed_gather_scatter(dev):
    from clipfs import ops, synth
    n, seq, width, vocab = 9, 16, 64, 512
    ids = synth.synth_captions(n, seq, vocab, seed=3, max_len=9)
    table, pos, ctx = _rand(vocab, width, seed=1), _rand(seq, width, seed=2), _rand(4, width, seed=3)
    D = lambda t: t.float().to(dev)
    x = ops.text_embed(ids.to(dev), D(table), D(pos))
    _close(x.reshape(n, seq, width), table[ids] + pos, 1e-6, "text embed")
    xc = ops.text_embed(ids.to(dev), D(table), D(pos), ctx=D(ctx))
    want = table[ids].clone()
    want[:, 1:5] = ctx
    _close(xc.reshape(n, seq, width), want + pos, 1e-6, "text embed ctx")
    rows, idx = ops.gather_eot(x, ids.to(dev))
    eot = ids.argmax(dim=-1)
    assert torch.equal(idx.cpu().long(), eot)
    _close(rows, (table[ids] + pos)[torch.arange(n), eot], 1e-6, "gather eot")
    dy = _rand(n, width, seed=5)
    dx = ops.scatter_rows(D(dy), idx, seq).reshape(n, seq, width)
    want = torch.zeros(n, seq, width, dtype=torch.float64)
    want[torch.arange(n), eot] = dy
    _close(dx, want, 1e-6, "scatter rows")
    dctx = torch.zeros(4, width, device=dev)
    g = _rand(n * seq, width, seed=6)
    ops.token_rows_grad(D(g), dctx, n, seq, 1)
    _close(dctx, g.reshape(n, seq, width)[:, 1:5].sum(0), 1e-5, "ctx grad")


def test_vit_fill_special(dev):
    from clipfs import ops
    B, P, nv, width = 3, 4, 4, 128
    L = 1 + P + nv
    cls, pos, vpt = _rand(width, seed=1), _rand(1 + P, width, seed=2), _rand(nv, width, seed=3)
    D = lambda t: t.float().to(dev)
    x = torch.zeros(B * L, width, device=dev)
    ops.vit_fill_special(x, D(cls), D(pos), D(vpt), B, L, P)
    x = x.reshape(B, L, width).cpu().double()
    assert torch.allclose(x[:, 0], (cls + pos[0]).float().double().expand(B, -1), atol=1e-6)
    assert torch.allclose(x[:, 1 + P:], vpt.float().double().expand(B, -1, -1), atol=1e-6)
    assert x[:, 1:1 + P].abs().max() == 0


def test_l2norm_classmean_ce_topk(dev):
    from clipfs import ops
    from oracle import clip_oracle as O
    D = lambda t: t.detach().float().to(dev)
    x = _rand(37, 512, seed=1).requires_grad_()
    y = O.l2_normalize(x)
    yg, inv = ops.l2norm_fwd(D(x), save_inv=True)
    _close(yg, y, 1e-6, "l2norm fwd")
    dy = _rand(37, 512, seed=2)
    y.backward(dy)
    _close(ops.l2norm_bwd(D(dy), yg, inv), x.grad, 1e-5, "l2norm bwd")
    # class mean over templates
    Cn, T, w = 11, 3, 128
    emb = _rand(Cn * T, w, seed=3).requires_grad_()
    cls_idx = [c for c in range(Cn) for _ in range(T)]
    want = O.class_text_features(emb, cls_idx, Cn).t()
    got = ops.class_mean_fwd(D(emb), Cn, T)
    _close(got, want, 1e-6, "class mean fwd")
    dout = _rand(Cn, w, seed=4)
    want.backward(dout)
    _close(ops.class_mean_bwd(D(emb), D(dout), Cn, T), emb.grad, 1e-5, "class mean bwd")
    # cross entropy
    logits = (_rand(50, 403, seed=5) * 5).requires_grad_()
    tgt = torch.from_numpy(np.random.RandomState(0).randint(0, 403, 50))
    loss = O.jt_cross_entropy(logits, tgt)
    loss.backward()
    ls, dl, correct = ops.cross_entropy(D(logits), tgt.to(dev))
    _close(ls / 50, loss.reshape(1), 1e-5, "ce loss")
    _close(dl, logits.grad, 1e-6, "ce grad")
    assert correct.item() == int((logits.argmax(1) == tgt).sum())
    # top-k with ties: smaller index first
    z = torch.tensor([[1.0, 3.0, 3.0, 2.0, 3.0, 0.5, 2.0], [5.0, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0]])
    lab = ops.topk(z.to(dev), 5).cpu().long()
    assert lab.tolist() == [[1, 2, 4, 3, 6], [0, 1, 2, 3, 4]]
    assert torch.equal(lab, O.jt_topk(z, 5))
    big = _rand(20, 403, seed=9).float()
    assert torch.equal(ops.topk(big.to(dev), 5).cpu().long(), O.jt_topk(big, 5))


def test_head_kernels(dev):
    from clipfs import ops
    from oracle import clip_oracle as O
    D = lambda t: t.float().to(dev)
    f = _rand(30, 512, seed=1)
    s1, b1 = 1 + 0.1 * _rand(512, seed=2), 0.1 * _rand(512, seed=3)
    w, b = _rand(403, 512, seed=4, scale=512 ** -0.5), 0.1 * _rand(403, seed=5)
    z = ops.gemm_nt(ops.channel_affine(D(f), D(s1), D(b1)), D(w), bias=D(b))
    want = O.channel_lp(f, s1, b1, w, b)
    _close(z, want, 1e-4, "channel_lp")
    _close(ops.logit_normalize(z), O.logit_normalize(want), 1e-4, "logit_normalize")


def test_adamw(dev):
    from clipfs import ops
    from oracle import clip_oracle as O
    n = 5000
    p, g = _rand(n, seed=1) * 0.05, _rand(n, seed=2) * 1e-3
    m, v = torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    pd, md, vd = p.float().to(dev), m.float().to(dev), v.float().to(dev)
    for step in (1, 2, 3):
        gs = g * step
        p, m, v = O.jt_adamw_step(p, gs, m, v, step)
        ops.adamw(pd, gs.float().to(dev), md, vd, step)
    _close(pd, p, 1e-6, "adamw p")
    _close(md, m, 1e-8, "adamw m")


def test_lora_bwd_from_f16_gradient_image(dev):
    """fp16 storage mode reads the incoming gradient of the adapter backward from its f16 image (clipfs_lora_bwd_f16dy): on
    an f16-exact dy it must give bitwise the results of the fp32 entry point (f16 -> f32 is exact, same kernels)."""
    from clipfs import _lib, ops
    rows, width, r, nseg, seed = 1000, 256, 16, 3, 0x77
    assert _lib.load().clipfs_lora_bwd_f16dy_ok(width, width, r, nseg) == 1
    assert _lib.load().clipfs_lora_bwd_f16dy_ok(192, 192, r, nseg) == 0  # outside the matrix-core kernels
    x = _rand(rows, width, seed=1).float().to(dev)
    A = _rand(nseg * r, width, seed=2, scale=width ** -0.5).float().to(dev)
    B = _rand(nseg * width, r, seed=3, scale=0.1).float().to(dev)
    dy16 = _rand(rows, nseg * width, seed=4).half().to(dev)
    t = ops.lora_down(x, A, r, nseg, p=0.25, seed=seed, stream_base=3)
    outs = []
    for dy in (dy16.float(), dy16):
        dA, dB, dx = torch.zeros_like(A), torch.zeros_like(B), torch.ones_like(x)
        dt = ops.lora_bwd(dy, x, t, A, B, dA, dB, dx=dx, scale=0.25, p=0.25, seed=seed, stream_base=3)
        outs.append((dt, dA, dB, dx))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][1].abs().max() > 0 and outs[0][2].abs().max() > 0


# ------------------------------------------------------------------ MTA
@pytest.mark.parametrize("V,d,Cn", [(65, 512, 403), (17, 64, 10)])
def test_mta(dev, V, d, Cn):
    from clipfs import ops
    from oracle import clip_oracle as O
    n_img = 3
    g = torch.Generator().manual_seed(4)
    base = torch.randn(n_img, 1, d, generator=g, dtype=torch.float64)
    feats = O.l2_normalize(base + 0.35 * torch.randn(n_img, V, d, generator=g, dtype=torch.float64))
    text = O.l2_normalize(torch.randn(Cn, d, generator=g, dtype=torch.float64) + 0.5 * base[0])
    mode, logits = ops.mta(feats.float().to(dev), text.float().to(dev))
    for i in range(n_img):
        # fp64 oracle on the kernel's own (fp32-exact) inputs; north-star tolerance 1e-3 on the 100 x cosine logits
        f64, t64 = feats[i].float().double(), text.float().double()
        wm = O.solve_mta(f64, t64.t(), return_mode=True)
        wl = O.solve_mta(f64, t64.t(), return_mode=False)
        _close(mode[i:i + 1], wm, 2e-5, "mta mode")
        _close(logits[i:i + 1], wl, 1e-3, "mta logits")
        assert torch.equal(ops.topk(logits[i:i + 1], 5).cpu().long(), O.jt_topk(wl.float(), 5))


def test_gemm_splitk_matches_unsplit(dev):
    """Few-tile shapes are cut along K (deterministic slab combine): same results as the unsplit kernel to
    fp32 rounding, epilogue (bias + LoRA + QuickGELU + residual) applied once after the combine."""
    from clipfs import _lib, ops
    from oracle import clip_oracle as O
    M, N, K, r = 300, 384, 1024, 4
    assert _lib.load().clipfs_gemm_splits(M, N, K) > 1
    a, w = _rand(M, K, seed=3), _rand(N, K, seed=4, scale=K ** -0.5)
    bias, res = _rand(N, seed=5), _rand(M, N, seed=6)
    t, lb = _rand(M, 3 * r, seed=7), _rand(N, r, seed=8)
    D = lambda x: x.float().to(dev)
    kw = dict(bias=D(bias), residual=D(res), lora_t=D(t), lora_b=D(lb), lora_seg_width=128, lora_scale=0.5, act=1)
    u1, u2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    y_split = ops.gemm_nt(D(a), D(w), aux_out=u1, **kw)
    y_plain = ops.gemm_nt(D(a), D(w), aux_out=u2, split_k=False, **kw)
    pre = a @ w.t() + bias
    for s in range(3):
        pre[:, s * 128:(s + 1) * 128] += 0.5 * t[:, s * r:(s + 1) * r] @ lb[s * 128:(s + 1) * 128].t()
    want = O.quick_gelu(pre) + res
    _close(y_split, want, 1e-4, "split-K gemm")
    _close(u1, pre, 1e-4, "split-K pre-activation")
    _close(y_split, y_plain.double(), 2e-5, "split vs unsplit")
    again = ops.gemm_nt(D(a), D(w), aux_out=u1, **kw)
    assert torch.equal(again, y_split), "split-K combine must be bitwise reproducible"


@pytest.mark.parametrize("M,N,K", [(200, 384, 128), (1000, 768, 768), (130, 403, 512)])
def test_gemm_bf16x3(dev, M, N, K):
    """Opt-in split-bf16 GEMM (3 bf16 MFMA products per operand pair): relative error ~1e-5 of the row/column
    norms -- between bf16 (4e-3) and exact fp32 (1e-7) -- and the fused epilogue is the fp32 one."""
    from clipfs import ops
    a, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    bias, res = _rand(N, seed=5), _rand(M, N, seed=6)
    D = lambda x: x.float().to(dev)
    planes = ops.split_bf16(D(w))
    hi = planes[0].view(torch.bfloat16).float().cpu().double()
    lo = planes[1].view(torch.bfloat16).float().cpu().double()
    w32 = w.float().double()
    assert (hi - w32.bfloat16().double()).abs().max() == 0           # hi = RNE bf16 of the weight
    assert (hi + lo - w32).abs().max() <= 2.0 ** -16 * w32.abs().max()    # two planes carry ~16 mantissa bits
    out = ops.gemm_nt(D(a), D(w), bias=D(bias), residual=D(res), b_planes=planes)
    want = a @ w.t() + bias + res
    err = (out.double().cpu() - want).abs().max().item()
    assert err < 6e-5, err
    exact = ops.gemm_nt(D(a), D(w), bias=D(bias), residual=D(res))
    assert (exact.double().cpu() - want).abs().max().item() < err * 2 + 1e-5  # sanity: fp32 path at least comparable


def test_gemm_f16_mode(dev):
    """fp16-operand MFMA kernel (cfg-5's "MFMA fp16 path"): operands rounded to f16, fp32 accumulate -- the result
    equals the fp64 product of the ROUNDED operands to fp32 accuracy; vs the unrounded product the error is that of
    fp16 rounding (2^-11 relative per operand)."""
    from clipfs import ops
    M, N, K = 300, 768, 1024
    a, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    D = lambda x: x.float().to(dev)
    plane = ops.to_f16(D(w))
    assert plane.dtype == torch.float16 and torch.equal(plane.cpu(), w.float().half())
    out = ops.gemm_nt(D(a), D(w), b_planes=plane)
    rounded = a.float().half().double() @ w.float().half().double().t()
    _close(out, rounded, 2e-5, "f16 gemm vs rounded operands")
    assert (out.double().cpu() - a @ w.t()).abs().max().item() < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,lora_r,act", [(300, 256, 64, 0, 0), (2056, 1024, 1024, 16, 0), (1100, 512, 2048, 4, 1),
                                               (33000, 1024, 256, 0, 2), (700, 384, 128, 0, 0),
                                               (16484, 2048, 128, 16, 1), (8192, 4096, 192, 4, 0), (20480, 1024, 1024, 16, 2),
                                               (16640, 1024, 2048, 16, 1), (16384, 2048, 1536, 4, 2)])
def test_gemm_f16_operands(M, N, K, lora_r, act):
    """f16 x f16 kernel (cfg-5 storage mode): both operands f16 in HBM, fp32 accumulate; f16 products are exact in
    fp32, so against an fp32 matmul of the same rounded operands only the summation order differs (tolerance 2e-5
    relative to the row scale); the f16 output copy is the fp32 result rounded once (2^-11 relative).
    The reference here is torch's matmul ON THE GPU (rocBLAS: an independent implementation, not the CPU oracle);
    test_gemm_f16_lds_epilogue_modes checks the same kernel against CPU fp64.  fp32 aux tensors / act 2 with a residual
    (this test) go through the register-epilogue kernel, f16 aux tensors (the tower) through the LDS epilogue."""
    from clipfs import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    a16, w16 = a.half(), ops.to_f16(w)
    kw = {}
    ref = a16.float() @ w16.float().T + bias
    if lora_r:
        segw = N // 2 if N % 256 == 0 else N
        nseg = N // segw
        t = torch.randn(M, nseg * lora_r, generator=g).cuda()
        lb = (torch.randn(N, lora_r, generator=g) * 0.1).cuda()
        kw = dict(lora_t=t, lora_b=lb, lora_seg_width=segw, lora_scale=0.25)
        t16 = t.half().float().view(M, nseg, lora_r)
        lb16 = (0.25 * lb).half().float()
        for sgi in range(nseg):
            ref[:, sgi * segw:(sgi + 1) * segw] += t16[:, sgi] @ lb16[sgi * segw:(sgi + 1) * segw].T
    aux_in = aux_out = pre = None
    if act == 1:
        aux_out = torch.empty(M, N, device="cuda")
        pre = ref.clone()
        ref = ref * torch.sigmoid(1.702 * ref)
    elif act == 2:
        aux_in = torch.randn(M, N, generator=g).cuda()
        sg = torch.sigmoid(1.702 * aux_in)
        ref = ref * (sg * (1 + 1.702 * aux_in * (1 - sg)))
    ref = ref + res
    out16 = torch.empty(M, N, device="cuda", dtype=torch.float16)
    out = ops.gemm_nt(None, w, bias=bias, residual=res, act=act, aux_out=aux_out, aux_in=aux_in, b_planes=w16, a16=a16,
                      out16=out16, **kw)
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() <= 2e-5 * scale + 1e-5
    assert (out16.float() - ref).abs().max().item() <= 1e-3 * scale
    if act == 1:
        assert (aux_out - pre).abs().max().item() <= 2e-5 * pre.abs().max().item() + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f16_bias", "f16_gelu_aux", "f16_gelu_grad", "f32_residual"])
@pytest.mark.parametrize("M,N,K", [(2048, 2560, 128), (2048 + 40, 2560, 192)])  # 80 tiles of 256 x 256: the phased kernel
def test_gemm_f16_lds_epilogue_modes(mode, M, N, K):
    """The 256 x 256 kernel's LDS epilogue in the four forms the fp16-storage tower uses (round 3 rewrote it: LDS-only
    barriers, loads of a pass before any store, 8 columns per thread for f16-only results): f16 result + bias (QKV),
    f16 result + QuickGELU + f16 pre-activation (c_fc), f16 result x QuickGELU'(f16 pre-activation) (d c_proj), fp32
    result + bias + residual (out / c_proj).  Reference: CPU fp64 on the same f16-rounded operands (NOT a GPU matmul);
    M = 2088 also runs the leftover-row kernel beside the big one."""
    from clipfs import ops
    g = torch.Generator().manual_seed(M + K)
    a16 = torch.randn(M, K, generator=g).half()
    w = torch.randn(N, K, generator=g) * K ** -0.5
    w16 = w.half()
    bias = torch.randn(N, generator=g)
    ref = a16.double() @ w16.double().T
    dev = torch.device("cuda:0")
    wd = w.to(dev)
    w16d = ops.to_f16(wd)
    assert torch.equal(w16d.cpu(), w16)
    kw = dict(b_planes=w16d, a16=a16.to(dev))
    if mode == "f32_residual":
        res = torch.randn(M, N, generator=g)
        out = ops.gemm_nt(None, wd, bias=bias.to(dev), residual=res.to(dev), **kw)
        want = ref + bias.double() + res.double()
        assert (out.double().cpu() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
        return
    out16 = torch.empty(M, N, device=dev, dtype=torch.float16)
    if mode == "f16_bias":
        ops.gemm_nt(None, wd, None, bias=bias.to(dev), out16=out16, only16=True, **kw)
        want = ref + bias.double()
    elif mode == "f16_gelu_aux":
        aux = torch.empty(M, N, device=dev, dtype=torch.float16)
        ops.gemm_nt(None, wd, None, bias=bias.to(dev), act=1, aux_out=aux, aux_f16=True, out16=out16, only16=True, **kw)
        pre = ref + bias.double()
        want = pre * torch.sigmoid(1.702 * pre)
        assert (aux.double().cpu() - pre).abs().max().item() <= 1e-3 * pre.abs().max().item()  # one f16 rounding
    else:
        u16 = torch.randn(M, N, generator=g).half()
        ops.gemm_nt(None, wd, None, act=2, aux_in=u16.to(dev), aux_f16=True, out16=out16, only16=True, **kw)
        sg = torch.sigmoid(1.702 * u16.double())
        want = ref * (sg * (1 + 1.702 * u16.double() * (1 - sg)))
    # one rounding to f16 of an fp32-accurate value (2^-11 relative to the element, bounded here by the tensor's scale)
    assert (out16.double().cpu() - want).abs().max().item() <= 1e-3 * want.abs().max().item()


@pytest.mark.gpu
def test_gemm_f16_phased_kernel_is_deterministic_and_matches_two_phase():
    """The 256 x 256 kernel with four phases per K-tile keeps four half-tiles of LDS-DMA in flight across its barriers:
    a misplaced wait would show as run-to-run differences or rare wrong tiles.  Ten launches of three shapes (K-tiles
    2, 3 and 16; 256 - 512 tiles) must be bitwise identical, and equal to the same f16 products summed in torch."""
    from clipfs import ops
    for M, N, K in ((16384, 4096, 128), (16384, 4096, 192), (32768, 1024, 1024), (16384, 1024, 2048)):  # last: 16x16x32
        g = torch.Generator().manual_seed(K)
        a16 = torch.randn(M, K, generator=g).cuda().half()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        w16 = ops.to_f16(w)
        first = ops.gemm_nt(None, w, b_planes=w16, a16=a16).clone()
        for _ in range(9):
            again = ops.gemm_nt(None, w, b_planes=w16, a16=a16)
            assert torch.equal(first, again), (M, N, K)
        for r0 in range(0, M, 8192):
            ref = a16[r0:r0 + 8192].float() @ w16.float().T
            assert (first[r0:r0 + 8192] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-5


@pytest.mark.gpu
def test_seq_row_helpers_and_eot_index():
    """One-row-per-sequence building blocks of clipfs_tower_fwd_rows / clipfs_tower_bwd_sparse through the C ABI: gather,
    put (other rows untouched), add, and the EOT position (argmax of the ids, first maximum: jclip/model.py:213-214)."""
    import ctypes as C
    from clipfs import _lib, ops
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    n, seq, width, ld = 37, 11, 20, 28
    src = torch.randn(n * seq, ld, generator=g).cuda()
    idx = torch.randint(0, seq, (n,), generator=g, dtype=torch.int32).cuda()
    flat = (torch.arange(n, device="cuda") * seq + idx.long())
    out = torch.empty(n, width, device="cuda")
    _lib.check(lib.clipfs_gather_seq_rows(src.data_ptr(), ld, idx.data_ptr(), out.data_ptr(), n, seq, width, st), "gather")
    assert torch.equal(out, src[flat, :width])
    dst = torch.randn(n * seq, ld, generator=g).cuda()
    want = dst.clone()
    want[flat, :width] = out
    _lib.check(lib.clipfs_put_seq_rows(out.data_ptr(), idx.data_ptr(), dst.data_ptr(), ld, n, seq, width, st), "put")
    assert torch.equal(dst, want)
    dx = torch.randn(n * seq, width, generator=g).cuda()
    want = dx.clone()
    want[flat] += out
    _lib.check(lib.clipfs_add_seq_rows(out.data_ptr(), idx.data_ptr(), dx.data_ptr(), n, seq, width, st), "add")
    assert torch.equal(dx, want)
    # EOT index: ties (two copies of the largest id) resolve to the first, as argmax does in the reference
    ids = torch.randint(1, 1000, (53, 77), generator=g, dtype=torch.int64)
    pos = torch.randint(1, 77, (53,), generator=g)
    ids[torch.arange(53), pos] = 49407
    ids[5, 70] = 49407
    ids[5, 3] = 49407
    got = ops.eot_index(ids.cuda())
    assert got.dtype == torch.int32 and torch.equal(got.cpu().long(), ids.argmax(dim=-1))
    assert got[5].item() == min(3, int(pos[5]))


def _attn_ref64(qkv, batch, seq, heads, causal=False):
    d = heads * 64
    x = qkv.double().view(batch, seq, 3, heads, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.full((seq, seq), float("-inf"), dtype=torch.float64, device=s.device).triu(1)
    lse = torch.logsumexp(s, -1)
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(batch * seq, d)
    return o, lse.reshape(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,seq,heads,causal", [(2, 257, 2, False), (1, 64, 1, False), (3, 50, 2, False),
                                                     (1, 288, 1, False), (2, 33, 3, True), (3, 77, 2, True),
                                                     (1, 130, 1, True)])
def test_attention_f16_fwd(batch, seq, heads, causal):
    """fp16-mode MFMA attention forward vs an fp64 reference on the SAME f16-rounded q, k, v: what remains is the f16
    rounding of the probabilities (2^-11 relative per term) and fp32 accumulation: 2e-3 absolute on outputs of
    unit scale, 1e-3 on the log-sum-exp."""
    from clipfs import ops
    g = torch.Generator().manual_seed(seq)
    qkv = torch.randn(batch * seq, 3 * heads * 64, generator=g).half().float().cuda()
    out, lse = ops.attention_f16_fwd(qkv, batch, seq, heads, causal)
    ro, rl = _attn_ref64(qkv, batch, seq, heads, causal)
    assert (out.double() - ro).abs().max().item() < 2e-3
    assert (lse.double() - rl).abs().max().item() < 1e-3
    # fp16 storage of qkv: the same values read as an f16 tensor give the same bits (the staging conversion is exact here)
    out_h, lse_h = ops.attention_f16_fwd(qkv.half(), batch, seq, heads, causal)
    assert torch.equal(out_h, out) and torch.equal(lse_h, lse)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,seq,heads,causal", [(2, 257, 2, False), (1, 64, 1, False), (2, 50, 2, False),
                                                     (1, 288, 1, False), (1, 100, 2, True), (3, 77, 2, True)])
def test_attention_f16_bwd(batch, seq, heads, causal):
    """fp16-mode MFMA attention backward vs fp64 autograd on the same f16-rounded inputs.  P, dS and the staged
    operands are rounded to f16 (2^-11 relative each), accumulation is fp32: 1e-2 relative to the largest gradient
    entry, stated here (the exact-fp32 kernels are held to 1e-5 in test_attention)."""
    from clipfs import ops
    g = torch.Generator().manual_seed(seq + 7)
    qkv = torch.randn(batch * seq, 3 * heads * 64, generator=g).half().float().cuda()
    dout = torch.randn(batch * seq, heads * 64, generator=g).half().float().cuda()
    out, lse = ops.attention_f16_fwd(qkv, batch, seq, heads, causal)
    dqkv = ops.attention_f16_bwd(qkv, dout, out, lse, batch, seq, heads, causal)
    assert torch.equal(ops.attention_f16_bwd(qkv.half(), dout, out, lse, batch, seq, heads, causal), dqkv)
    # dO handed over as its f16 image (fp16 storage mode: the output-projection dgrad writes nothing else): same bits,
    # with either storage of qkv -- dout above is f16-representable and the fp32 staging rounds it the same way
    assert torch.equal(ops.attention_f16_bwd(qkv, dout.half(), out, lse, batch, seq, heads, causal), dqkv)
    assert torch.equal(ops.attention_f16_bwd(qkv.half(), dout.half(), out, lse, batch, seq, heads, causal), dqkv)
    x = qkv.double().requires_grad_(True)
    ro, _ = _attn_ref64(x, batch, seq, heads, causal)
    (ro * dout.double()).sum().backward()
    ref = x.grad
    assert (dqkv.double() - ref).abs().max().item() < 1e-2 * ref.abs().max().item()


def test_stream_k_gemm_schedule_in_subprocess():
    """The opt-in stream-K schedule (CLIPFS_GEMM_SK=2: persistent workgroups, partial tiles summed by the last arriver in
    run order) is read from the environment once per process, so it is exercised in a child: results within fp32
    rounding of an fp64 product, bitwise reproducible from call to call, epilogue options and ragged shapes included."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from clipfs import ops, _lib
assert _lib.load().clipfs_gemm_counter_ints(1600, 768, 768) > 0, "stream-K is not on"
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
for M, N, K in ((1600, 768, 768), (12800, 768, 3072), (1000, 200, 96), (77, 512, 2048), (3927, 1536, 512), (4096, 4096, 1024)):
    a = torch.randn(M, K, generator=g).to(dev); b = torch.randn(N, K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev); res = torch.randn(M, N, generator=g).to(dev)
    o1 = ops.gemm_nt(a, b, bias=bias, residual=res, act=1, aux_out=torch.empty(M, N, device=dev))
    o2 = ops.gemm_nt(a, b, bias=bias, residual=res, act=1, aux_out=torch.empty(M, N, device=dev))
    u = a.double() @ b.double().t() + bias.double()
    want = u * torch.sigmoid(1.702 * u) + res.double()
    err = (o1.double() - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-5, (M, N, K, err)
    assert torch.equal(o1, o2), (M, N, K)
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CLIPFS_GEMM_SK="2")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(root, "jittor-clip-fewshot_amd")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
