"""CPU oracle for the CLIP few-shot hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, in plain PyTorch-CPU tensor algebra, of
the arithmetic the reference (Dokumushikun/jittor-clip-fewshot, Jittor 1.3.8.5)
performs on the path named by BASELINE.json:north_star.  It exists to CHECK the
HIP engine; nothing under ``jittor-clip-fewshot_amd/`` may import it.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it.

PARITY STATUS: **parity unpinned against Jittor itself.**  Jittor is not
installable in the build container (no network) and the reference ships no
tests / golden vectors.  What pins this oracle:
  * the reference's own fixtures: ``lora_weights1/lora_weights.pkl`` (real
    trained LoRA A/B), ``jclip/bpe_simple_vocab_16e6.txt`` (BPE merges);
  * an independent second opinion: ``transformers.CLIPModel`` (OpenAI CLIP
    semantics) with locally initialised random weights
    (tests/test_oracle_vs_hf.py, CPU only);
  * known-answer tests (Random123 Philox4x32-10 vectors, published CLIP token ids)
    and algebraic self checks (merged LoRA == additive LoRA at p=0, packed QKV ==
    split q/k/v, caption chunking invariance, ...) in tests/test_oracle_golden.py.
Jittor op semantics that cannot be verified offline are isolated in single,
clearly named functions (``jt_layer_norm``, ``jt_cross_entropy``, ``jt_adamw_step``,
``jt_std``, ``jt_argsort_values``, ``jt_topk``).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  All functions are dtype-generic: run them in float64 for
"truth" and in float32 for "what the reference would produce".
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# Jittor-semantics primitives (each isolated: parity unpinned, see header)
# --------------------------------------------------------------------------


def jt_layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """jittor.nn.LayerNorm as used by jclip/model.py:17-21: biased variance,
    eps inside the sqrt, affine."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * weight + bias


def quick_gelu(x: Tensor) -> Tensor:
    """jclip/model.py:24-27."""
    return x * torch.sigmoid(1.702 * x)


def jt_linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """jittor.nn.linear: x @ w.T + b   (w is [out, in])."""
    y = x @ w.t()
    return y if b is None else y + b


def jt_softmax(x: Tensor, dim: int = -1) -> Tensor:
    """jittor.nn.softmax: max-subtracted exp / sum."""
    x = x - x.max(dim=dim, keepdim=True).values
    e = torch.exp(x)
    return e / e.sum(dim=dim, keepdim=True)


def jt_cross_entropy(logits: Tensor, target: Tensor) -> Tensor:
    """jittor.nn.cross_entropy_loss (lora_train_vlp.py:997): mean over the batch of
    logsumexp(row) - row[target]."""
    z = logits - logits.max(dim=1, keepdim=True).values
    lse = torch.log(torch.exp(z).sum(dim=1))
    picked = z.gather(1, target.view(-1, 1).long()).squeeze(1)
    return (lse - picked).mean()


def jt_std(x: Tensor) -> Tensor:
    """jittor.std over ALL elements: unbiased (n-1), variance clamped at 1e-6
    (slow_pace.py:1277)."""
    n = x.numel()
    var = ((x - x.mean()) ** 2).sum() / (n - 1)
    return torch.sqrt(torch.clamp(var, min=1e-6))


def jt_argsort_values(x: Tensor, dim: int = 1) -> Tensor:
    """``_, sorted = jt.argsort(x, dim)``: Jittor's argsort returns
    (indices, sorted values); the reference keeps the VALUES
    (lora_train_vlp.py:755)."""
    return torch.sort(x, dim=dim).values


def jt_topk(x: Tensor, k: int) -> Tensor:
    """``x.topk(k, 1, True, True)[1]`` (lora_train_vlp.py:639).  Tie order is
    not specified by Jittor; the build fixes it: larger value first, on ties the
    SMALLER class index first (stable)."""
    # stable descending sort == argsort of (-x) with a stable algorithm
    idx = torch.sort(-x, dim=1, stable=True).indices
    return idx[:, :k]


def jt_adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float = 2e-4,
                  betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                  weight_decay: float = 1e-2) -> Tuple[Tensor, Tensor, Tensor]:
    """jittor.optim.AdamW.step (lora_train_vlp.py:946,1002): decoupled decay
    ``p *= 1 - lr*wd`` first, then bias-corrected Adam with
    ``denom = sqrt(v)/sqrt(bc2) + eps``.  Returns (p, m, v) new values."""
    b0, b1 = betas
    p = p * (1 - lr * weight_decay)
    m = b0 * m + (1 - b0) * g
    v = b1 * v + (1 - b1) * g * g
    bc1 = 1 - b0 ** step
    bc2 = 1 - b1 ** step
    denom = torch.sqrt(v) / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


# --------------------------------------------------------------------------
# Counter based dropout mask (Philox4x32-10).  The reference uses
# jittor.nn.Dropout whose RNG stream cannot be reproduced; the build defines
# its own stream so that GPU and oracle masks are bit-identical.
# --------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32_10(ctr: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """Philox4x32-10 (Salmon et al. 2011).  ``ctr`` uint32 [n,4] -> uint32 [n,4]."""
    c = ctr.astype(np.uint32).copy()
    k0 = np.uint32(key[0])
    k1 = np.uint32(key[1])
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c[:, 0].astype(np.uint64) * _PHILOX_M0
            p1 = c[:, 2].astype(np.uint64) * _PHILOX_M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = p1.astype(np.uint32)
            n0 = hi1 ^ c[:, 1] ^ k0
            n1 = lo1
            n2 = hi0 ^ c[:, 3] ^ k1
            n3 = lo0
            c = np.stack([n0, n1, n2, n3], axis=1)
            k0 = np.uint32(k0 + _PHILOX_W0)
            k1 = np.uint32(k1 + _PHILOX_W1)
    return c


def dropout_keep_mask(seed: int, stream: int, rows: int, cols: int, p: float) -> np.ndarray:
    """Keep mask (bool [rows, cols]) of the build's dropout: element (r, c) is
    generated by Philox counter (c//4, r, stream, 0), key (seed_lo, seed_hi),
    lane c%4; kept iff  u32 >= floor(p * 2^32)."""
    assert cols % 4 == 0
    thr = np.uint32(min(int(p * 4294967296.0), 4294967295))
    r_idx, q_idx = np.meshgrid(np.arange(rows, dtype=np.uint32),
                               np.arange(cols // 4, dtype=np.uint32), indexing="ij")
    ctr = np.stack([q_idx.ravel(), r_idx.ravel(),
                    np.full(r_idx.size, stream, np.uint32),
                    np.zeros(r_idx.size, np.uint32)], axis=1)
    out = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return (out >= thr).reshape(rows, cols)


# --------------------------------------------------------------------------
# LoRA linear  (lora_train_vlp.py:185-306)
# --------------------------------------------------------------------------


def lora_scaling(alpha: float, r: int) -> float:
    """lora_train_vlp.py:197  --  alpha / sqrt(r)  (NOT alpha / r)."""
    return alpha / math.sqrt(r)


def lora_linear(x: Tensor, w: Tensor, b: Optional[Tensor], lora_a: Optional[Tensor],
                lora_b: Optional[Tensor], scaling: float,
                drop_scale: Optional[Tensor] = None) -> Tensor:
    """LinearLoRA.execute, dropout branch (lora_train_vlp.py:296-306), which is the
    branch that always runs with the default ``dropout_rate=0.25`` because
    Jittor's Module.train()/eval() never call LinearLoRA.train:

        y = x W^T + b + scaling * ( drop(x) @ (B @ A)^T )

    ``drop_scale`` is the dropout multiplier (0 or 1/(1-p)) with x's shape, or
    None (eval / p = 0).  Forms B@A exactly like merge_BA (:218-221)."""
    y = jt_linear(x, w, b)
    if lora_a is None:
        return y
    xd = x if drop_scale is None else x * drop_scale
    ba = lora_b @ lora_a  # [out, in]
    return y + (xd @ ba.t()) * scaling


def lora_linear_merged(x: Tensor, w: Tensor, b: Optional[Tensor], lora_a: Tensor, lora_b: Tensor,
                       scaling: float) -> Tensor:
    """No-dropout branch (lora_train_vlp.py:287-294): W <- W + s*BA, linear."""
    return jt_linear(x, w + (lora_b @ lora_a) * scaling, b)


# --------------------------------------------------------------------------
# Attention  (jclip/mha.py:55-83,129-146,437-466 ; lora_train_vlp.py:339-367,431-506)
# --------------------------------------------------------------------------


def sdpa(q: Tensor, k: Tensor, v: Tensor, attn_mask: Optional[Tensor]) -> Tensor:
    """scaled_dot_product_attention, mha.py:55-83: scale applied AFTER q@k^T,
    additive float mask, softmax over keys, no dropout (p = 0)."""
    scale = 1.0 / math.sqrt(q.shape[-1])
    w = (q @ k.transpose(-2, -1)) * scale
    if attn_mask is not None:
        w = w + attn_mask
    w = jt_softmax(w, dim=-1)
    return w @ v


def build_causal_mask(n_ctx: int, dtype=torch.float32) -> Tensor:
    """CLIP.build_attention_mask, model.py:189-193: -inf strictly above the diagonal."""
    m = torch.full((n_ctx, n_ctx), float("-inf"), dtype=dtype)
    return torch.triu(m, diagonal=1)


def mha_forward(x: Tensor, blk: Dict[str, Tensor], heads: int, attn_mask: Optional[Tensor],
                lora: Optional[dict] = None, scaling: float = 0.0,
                drop: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """Self-attention on a sequence-first tensor x[L, N, d].

    Un-adapted: multi_head_attention_forward need_weights=False branch,
    mha.py:437-466 with _in_projection_packed (:129-146): packed linear then split
    in (q, k, v) order.  Adapted: PlainMultiheadAttentionLoRA.forward_module,
    lora_train_vlp.py:431-506 with in_proj rows [0:d],[d:2d],[2d:3d] as
    q/k/v weights (:395-409).  ``lora`` = {'q_proj': {'w_lora_A','w_lora_B'}, ...}
    (keys present only for adapted projections; 'proj' = out projection).
    ``drop`` = per projection dropout multipliers (same keys) or None."""
    L, N, d = x.shape
    hd = d // heads
    w_in, b_in = blk["in_proj_weight"], blk["in_proj_bias"]
    outs = []
    for i, name in enumerate(("q_proj", "k_proj", "v_proj")):
        w = w_in[i * d:(i + 1) * d]
        b = b_in[i * d:(i + 1) * d]
        ad = (lora or {}).get(name)
        if ad is not None:
            ds = None if drop is None else drop.get(name)
            outs.append(lora_linear(x, w, b, ad["w_lora_A"], ad["w_lora_B"], scaling, ds))
        else:
            outs.append(jt_linear(x, w, b))
    q, k, v = outs
    # [L, N*H, hd] -> [N*H, L, hd] -> [N, H, L, hd]   (mha.py:439-446 / lora:492-498)
    q = q.reshape(L, N * heads, hd).transpose(0, 1).reshape(N, heads, L, hd)
    k = k.reshape(L, N * heads, hd).transpose(0, 1).reshape(N, heads, L, hd)
    v = v.reshape(L, N * heads, hd).transpose(0, 1).reshape(N, heads, L, hd)
    o = sdpa(q, k, v, attn_mask)
    o = o.permute(2, 0, 1, 3).reshape(L * N, d)  # mha.py:458 / lora:501
    ad = (lora or {}).get("proj")
    if ad is not None:
        ds = None if drop is None else drop.get("proj")
        if ds is not None:
            ds = ds.reshape(L * N, d)
        o = lora_linear(o, blk["out_proj.weight"], blk["out_proj.bias"], ad["w_lora_A"], ad["w_lora_B"],
                        scaling, ds)
    else:
        o = jt_linear(o, blk["out_proj.weight"], blk["out_proj.bias"])
    return o.reshape(L, N, d)


def resblock_forward(x: Tensor, blk: Dict[str, Tensor], heads: int, attn_mask: Optional[Tensor],
                     lora: Optional[dict] = None, scaling: float = 0.0,
                     drop: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """ResidualAttentionBlock.execute, model.py:59-62 (pre-LN)."""
    h = jt_layer_norm(x, blk["ln_1.weight"], blk["ln_1.bias"])
    x = x + mha_forward(h, blk, heads, attn_mask, lora, scaling, drop)
    h = jt_layer_norm(x, blk["ln_2.weight"], blk["ln_2.bias"])
    h = jt_linear(h, blk["mlp.c_fc.weight"], blk["mlp.c_fc.bias"])
    h = quick_gelu(h)
    h = jt_linear(h, blk["mlp.c_proj.weight"], blk["mlp.c_proj.bias"])
    return x + h


def _block_params(sd: Dict[str, Tensor], prefix: str, i: int) -> Dict[str, Tensor]:
    p = f"{prefix}.resblocks.{i}."
    names = ("ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
             "mlp.c_proj.weight", "mlp.c_proj.bias")
    out = {n: sd[p + n] for n in names}
    out["in_proj_weight"] = sd[p + "attn.in_proj_weight"]
    out["in_proj_bias"] = sd[p + "attn.in_proj_bias"]
    out["out_proj.weight"] = sd[p + "attn.out_proj.weight"]
    out["out_proj.bias"] = sd[p + "attn.out_proj.bias"]
    return out


def count_layers(sd: Dict[str, Tensor], prefix: str) -> int:
    """build_model, model.py:240-243,271-274."""
    return len({k.split(".resblocks.")[1].split(".")[0] for k in sd if k.startswith(prefix + ".resblocks.")})


def transformer_forward(x: Tensor, sd: Dict[str, Tensor], prefix: str, heads: int,
                        attn_mask: Optional[Tensor], lora_layers: Optional[Dict[int, dict]] = None,
                        scaling: float = 0.0, drops: Optional[Dict[int, Dict[str, Tensor]]] = None) -> Tensor:
    """Transformer.execute, model.py:65-77.  ``lora_layers`` maps block index ->
    adapter dict; ``drops`` maps block index -> per projection multipliers."""
    for i in range(count_layers(sd, prefix)):
        x = resblock_forward(x, _block_params(sd, prefix, i), heads, attn_mask,
                             None if lora_layers is None else lora_layers.get(i), scaling,
                             None if drops is None else drops.get(i))
    return x


# --------------------------------------------------------------------------
# Towers  (jclip/model.py:80-126,199-232 ; jclip/model1.py:160-207 ; slow_pace.py:837-848)
# --------------------------------------------------------------------------


def encode_image(sd: Dict[str, Tensor], image: Tensor, lora_layers: Optional[Dict[int, dict]] = None,
                 scaling: float = 0.0, vpt: Optional[Tensor] = None,
                 drops: Optional[Dict[int, Dict[str, Tensor]]] = None,
                 return_tokens: bool = False) -> Tensor:
    """VisionTransformer.execute, model.py:104-126; with ``vpt`` [n_ctx, width]
    the shallow-VPT variant model1.py:180-207 (tokens appended AFTER the patch
    tokens and after the positional embedding, before ln_pre)."""
    w = sd["visual.conv1.weight"]
    width, _, ps, _ = w.shape
    x = torch.nn.functional.conv2d(image, w, bias=None, stride=ps)  # [B, width, g, g]
    B = x.shape[0]
    x = x.reshape(B, width, -1).permute(0, 2, 1)  # [B, g*g, width]
    cls = sd["visual.class_embedding"].to(x.dtype) + torch.zeros(B, 1, width, dtype=x.dtype)
    x = torch.cat([cls, x], dim=1)
    x = x + sd["visual.positional_embedding"].to(x.dtype)
    if vpt is not None:
        x = torch.cat([x, vpt.unsqueeze(0).expand(B, -1, -1)], dim=1)  # model1.py:192-194
    x = jt_layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    x = x.permute(1, 0, 2)  # NLD -> LND
    x = transformer_forward(x, sd, "visual.transformer", width // 64, None, lora_layers, scaling, drops)
    x = x.permute(1, 0, 2)
    if return_tokens:
        return x
    x = jt_layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return x @ sd["visual.proj"]


def encode_text(sd: Dict[str, Tensor], text: Tensor, lora_layers: Optional[Dict[int, dict]] = None,
                scaling: float = 0.0, embeds: Optional[Tensor] = None,
                drops: Optional[Dict[int, Dict[str, Tensor]]] = None) -> Tensor:
    """CLIP.encode_text, model.py:202-215.  With ``embeds`` [N, 77, d] given the
    token embedding lookup is skipped: TextEncoder.execute, slow_pace.py:837-848
    (``text`` is then only used to locate the EOT row = argmax of the ids)."""
    x = sd["token_embedding.weight"][text.long()] if embeds is None else embeds
    x = x + sd["positional_embedding"]
    n_ctx = x.shape[1]
    width = x.shape[2]
    x = x.permute(1, 0, 2)
    mask = build_causal_mask(n_ctx, x.dtype)
    x = transformer_forward(x, sd, "transformer", width // 64, mask, lora_layers, scaling, drops)
    x = x.permute(1, 0, 2)
    x = jt_layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    eot = text.argmax(dim=-1)  # Jittor argmax returns (index, value); [0] = index (model.py:214)
    x = x[torch.arange(x.shape[0]), eot] @ sd["text_projection"]
    return x


def l2_normalize(x: Tensor) -> Tensor:
    """x / x.norm(dim=-1, keepdim=True)   (model.py:222-224, lora_train_vlp.py:993)."""
    return x / torch.sqrt((x * x).sum(dim=-1, keepdim=True))


def clip_forward(sd: Dict[str, Tensor], image: Tensor, text: Tensor) -> Tuple[Tensor, Tensor]:
    """CLIP.execute, model.py:217-232 (uses logit_scale.exp())."""
    fi = l2_normalize(encode_image(sd, image))
    ft = l2_normalize(encode_text(sd, text))
    li = sd["logit_scale"].exp() * fi @ ft.t()
    return li, li.t()


# --------------------------------------------------------------------------
# LoRA placement / checkpoint schema  (lora_train_vlp.py:27-63,516-635)
# --------------------------------------------------------------------------

INDEX_POSITIONS_TEXT = {
    "top1": [11], "top2": [10, 11], "top3": [9, 10, 11], "bottom": [0, 1, 2, 3], "mid": [4, 5, 6, 7],
    "up": [8, 9, 10, 11], "half-up": [6, 7, 8, 9, 10, 11], "half-bottom": [0, 1, 2, 3, 4, 5],
    "all": list(range(12)),
}
INDEX_POSITIONS_VISION = {
    "ViT-B/16": {"top": [11], "top3": [9, 10, 11], "bottom": [0, 1, 2, 3], "mid": [4, 5, 6, 7],
                 "up": [8, 9, 10, 11], "half-up": [6, 7, 8, 9, 10, 11], "half-bottom": [0, 1, 2, 3, 4, 5],
                 "all": list(range(12))},
    "ViT-B/32": {"bottom": [0, 1, 2, 3], "mid": [4, 5, 6, 7], "up": [8, 9, 10, 11],
                 "half-up": [6, 7, 8, 9, 10, 11], "half-bottom": [0, 1, 2, 3, 4, 5], "all": list(range(12))},
    # quirk kept: 'all' covers blocks 0..20 of 24 only (lora_train_vlp.py:62)
    "ViT-L/14": {"bottom": [0, 1, 2, 3], "mid": [4, 5, 6, 7], "up": [8, 9, 10, 11],
                 "half-up": [6, 7, 8, 9, 10, 11], "half-bottom": [0, 1, 2, 3, 4, 5], "all": list(range(21))},
}


def split_lora_checkpoint(weights: Dict[str, dict], encoder: str, position: str, backbone: str,
                          dtype=torch.float64) -> Tuple[Dict[int, dict], Dict[int, dict]]:
    """Map the flat ``layer_i`` list of save_lora (lora_train_vlp.py:551-593) back to
    (text {block: adapters}, vision {block: adapters}) following apply_lora's
    traversal order: text blocks first, then vision (:516-548)."""
    text, vis = {}, {}
    i = 0
    conv = lambda d: {p: {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in ab.items()} for p, ab in d.items()}
    if encoder in ("text", "both"):
        for blk in INDEX_POSITIONS_TEXT[position]:
            text[blk] = conv(weights[f"layer_{i}"])
            i += 1
    if encoder in ("vision", "both"):
        for blk in INDEX_POSITIONS_VISION[backbone][position]:
            vis[blk] = conv(weights[f"layer_{i}"])
            i += 1
    return text, vis


def kaiming_uniform_a5(rows: int, cols: int, gen: torch.Generator, dtype=torch.float32) -> Tensor:
    """nn.init.kaiming_uniform_(A, a=sqrt(5)) on A[r, in] (lora_train_vlp.py:212):
    bound = sqrt(6 / ((1 + a^2) * fan_in)) = 1/sqrt(in)."""
    bound = 1.0 / math.sqrt(cols)
    return ((torch.rand(rows, cols, generator=gen, dtype=torch.float64) * 2 - 1) * bound).to(dtype)


# --------------------------------------------------------------------------
# Classifier assembly, loss  (lora_train_vlp.py:647-666,964-997)
# --------------------------------------------------------------------------


def class_text_features(emb: Tensor, class_index: Sequence[int], n_classes: int) -> Tensor:
    """In-loop classifier assembly, lora_train_vlp.py:978-990 (== clip_classifier
    :647-666): per class, L2-normalise each template embedding, mean over the
    class's templates, L2-normalise again; stacked -> [d, C]."""
    cols = []
    idx = torch.as_tensor(list(class_index))
    for c in range(n_classes):
        e = emb[idx == c]
        e = l2_normalize(e)
        m = e.mean(dim=0)
        cols.append(m / torch.sqrt((m * m).sum()))
    return torch.stack(cols, dim=1)


def train_logits(img_feat: Tensor, txt_feat_dc: Tensor, logit_scale: float = 100.0) -> Tensor:
    """lora_train_vlp.py:993-995: ``logit_scale * norm(img) @ textual_features``
    (left-to-right: the scale multiplies the image features first)."""
    return (logit_scale * l2_normalize(img_feat)) @ txt_feat_dc


def cls_acc(output: Tensor, target: Tensor, topk: int = 1) -> float:
    """lora_train_vlp.py:638-644."""
    pred = jt_topk(output, topk).t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return 100.0 * float(correct[:topk].reshape(-1).double().sum()) / target.shape[0]


def cls_acc_ood(output: Tensor, target: Tensor, topk: int = 1) -> float:
    """ood.py:638-652: correct iff prediction and label fall on the same side of
    class 373 (quirk kept: ``< 373`` here while split_ood uses ``<= 372`` -- the
    same set -- and base classes are 0..373 in classes.txt)."""
    pred = jt_topk(output, topk).t()
    t = target.view(1, -1)
    correct = ((pred < 373) & (t < 373)) | ((pred >= 373) & (t >= 373))
    return 100.0 * float(correct[:topk].reshape(-1).double().sum()) / target.shape[0]


def ood_is_base(mta_logits: Tensor) -> Tensor:
    """split_ood, ood.py:877-883: argmax <= 372 -> base list, else new list."""
    return mta_logits.argmax(dim=-1) <= 372


# --------------------------------------------------------------------------
# Prompt learner / LP++-style head  (slow_pace.py:110-205,1195-1206,1276-1280)
# --------------------------------------------------------------------------


def build_prompts(ctx: Tensor, token_embedding: Tensor, tokenized_prompts: Tensor) -> Tensor:
    """VLPromptLearner.execute, slow_pace.py:172-173,185-199: per class
    [SOT embedding | ctx (shared, n_ctx rows) | embedding of the remaining tokens]."""
    emb = token_embedding[tokenized_prompts.long()]  # [C, 77, d]
    n_ctx = ctx.shape[0]
    C = emb.shape[0]
    return torch.cat([emb[:, :1, :], ctx.unsqueeze(0).expand(C, -1, -1), emb[:, 1 + n_ctx:, :]], dim=1)


def channel_lp(features: Tensor, scale1: Tensor, bias1: Tensor, fc_w: Tensor, fc_b: Tensor) -> Tensor:
    """Channel_LP.execute, slow_pace.py:1195-1206."""
    return jt_linear(scale1.unsqueeze(0) * features + bias1.unsqueeze(0), fc_w, fc_b)


def logit_normalize(logit: Tensor) -> Tensor:
    """slow_pace.py:1276-1280: (z - rowmean) / std over ALL elements."""
    return (logit - logit.mean(dim=1, keepdim=True)) / jt_std(logit)


# --------------------------------------------------------------------------
# Stage-2 objective (slow_pace.py:1590-1697), MoCo branch excluded
# --------------------------------------------------------------------------


def jt_l1_loss(output: Tensor, target: Tensor) -> Tensor:
    """jittor.nn.l1_loss: mean |output - target| (slow_pace.py:1654-1655)."""
    return (output - target).abs().mean()


def kl_div(log_probs: Tensor, target_log_probs: Tensor, reduction: str = "sum") -> Tensor:
    """slow_pace.py:1170-1177 verbatim semantics: exp(target) * (target - log_probs)."""
    kl = torch.exp(target_log_probs) * (target_log_probs - log_probs)
    if reduction == "sum":
        return kl.sum()
    if reduction == "mean":
        return kl.mean()
    return kl


def jt_log_softmax(x: Tensor, dim: int = 1) -> Tensor:
    z = x - x.max(dim=dim, keepdim=True).values
    return z - torch.log(torch.exp(z).sum(dim=dim, keepdim=True))


def scl_logits_loss(cosine_similarity: Tensor, zero_shot_logits: Tensor) -> Tensor:
    """slow_pace.py:1656-1658."""
    a = jt_log_softmax(cosine_similarity / 1, 1)
    b = jt_log_softmax(zero_shot_logits / 1, 1)
    return kl_div(a, b, reduction="sum") * (1 * 1) / cosine_similarity.numel()


def cosine_annealing_lr(base_lr: float, t: int, T_max: int, eta_min: float = 1e-6) -> float:
    """jt.lr_scheduler.CosineAnnealingLR (slow_pace.py:1591): lr after t scheduler steps (closed form)."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * t / T_max)) / 2.0


def stage2_loss(image_features: Tensor, text_features: Tensor, target: Tensor, zs_image_features: Tensor,
                zs_text_features: Tensor, lp_params: Tuple[Tensor, Tensor, Tensor, Tensor], lp_image_features: Tensor,
                lp_text_features: Tensor) -> Tuple[Tensor, Dict[str, Tensor], Tensor]:
    """slow_pace.py:1636-1688 without loss_aux (MoCo): sim_ce + (L_SCL_logits + L1 text + L1 image) + lp_ce.
    ``lp_params`` = (scale1, bias1, fc.weight, fc.bias) of Channel_LP."""
    img = image_features / image_features.norm(dim=-1, keepdim=True)            # :1630
    txt = text_features / text_features.norm(dim=-1, keepdim=True)              # :1631
    cos = 100 * img @ txt.t()                                                   # :1640
    zs_logits = 100 * zs_image_features @ zs_text_features.t()                  # :1650
    scl_text = jt_l1_loss(txt, zs_text_features)                                # :1654
    scl_image = jt_l1_loss(img, zs_image_features)                              # :1655
    scl_logits = scl_logits_loss(cos, zs_logits)                                # :1656-1658
    feats = torch.cat((lp_image_features, lp_text_features), dim=0)             # :1663
    out_lp = logit_normalize(channel_lp(feats, *lp_params))                     # :1665-1666
    C = text_features.shape[0]
    tgt = torch.cat((target.long(), torch.arange(C)))                           # :1667-1668
    lp_ce = jt_cross_entropy(out_lp, tgt)                                       # :1669
    sim_ce = jt_cross_entropy(cos, target)                                      # :1686
    loss = sim_ce + (scl_logits + scl_text + scl_image) + lp_ce                 # :1684,1688 (minus loss_aux)
    return loss, {"sim_ce": sim_ce, "scl_text": scl_text, "scl_image": scl_image, "scl_logits": scl_logits,
                  "lp_ce": lp_ce}, cos


# --------------------------------------------------------------------------
# MoCo-v3 ResNet-50 auxiliary branch  (slow_pace.py:1208-1219,1237-1271,1677-1680)
# --------------------------------------------------------------------------


def resnet50_forward(sd: Dict[str, Tensor], x: Tensor, eps: float = 1e-5) -> Tensor:
    """jittor.models.resnet.resnet50 (= torchvision ResNet-50 v1.5: Bottleneck with the stride on the 3x3 convolution)
    with ``fc`` replaced by Identity (slow_pace.py:1269), inference-mode BatchNorm (running statistics; the build's
    flagged choice, see clipfs/resnet.py): x [B, 3, H, W] normalised images -> [B, 2048] pooled features."""
    import torch.nn.functional as F

    def bn(t, name):
        return F.batch_norm(t, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                            False, 0.0, eps)

    y = F.relu(bn(F.conv2d(x, sd["conv1.weight"], None, 2, 3), "bn1"))
    y = F.max_pool2d(y, 3, 2, 1)
    for li, nb in enumerate((3, 4, 6, 3)):
        for bi in range(nb):
            p = f"layer{li + 1}.{bi}"
            stride = 2 if (bi == 0 and li > 0) else 1
            idn = y
            o = F.relu(bn(F.conv2d(y, sd[p + ".conv1.weight"]), p + ".bn1"))
            o = F.relu(bn(F.conv2d(o, sd[p + ".conv2.weight"], None, stride, 1), p + ".bn2"))
            o = bn(F.conv2d(o, sd[p + ".conv3.weight"]), p + ".bn3")
            if bi == 0:
                idn = bn(F.conv2d(y, sd[p + ".downsample.0.weight"], None, stride), p + ".downsample.1")
            y = F.relu(o + idn)
    return y.mean(dim=(2, 3))


def moco_aux_loss(features: Tensor, fc_w: Tensor, fc_b: Tensor, target: Tensor) -> Tensor:
    """slow_pace.py:1678-1680: CE(logit_normalize(Moco_Adapter(features)), target)."""
    return jt_cross_entropy(logit_normalize(jt_linear(features, fc_w, fc_b)), target.long())


# --------------------------------------------------------------------------
# MTA  (lora_train_vlp.py:733-811 ; slow_pace.py:1363-1433)
# --------------------------------------------------------------------------


def gaussian_kernel(mu: Tensor, bandwidth: Tensor, datapoints: Tensor) -> Tensor:
    """lora_train_vlp.py:733-736."""
    dist = torch.sqrt(((datapoints - mu) ** 2).sum(dim=-1))
    return torch.exp(-dist ** 2 / (2 * bandwidth ** 2))


def cdist(x1: Tensor, x2: Tensor) -> Tensor:
    """lora_train_vlp.py:737-741: sqrt(|x1|^2 - 2 x1 x2^T + |x2|^2).
    DEVIATION (flagged): the radicand is clamped at 0.  In fp32 the reference's
    diagonal can round to a tiny negative number and sqrt gives NaN, whose sort
    position in Jittor is unspecified; exact arithmetic gives 0."""
    a = (x1 ** 2).sum(dim=1, keepdim=True)
    b = (x2 ** 2).sum(dim=1, keepdim=True)
    return torch.sqrt(torch.clamp(a - 2 * (x1 @ x2.t()) + b.t(), min=0))


def solve_mta(image_features: Tensor, text_features: Tensor, return_mode: bool = False,
              return_trace: bool = False):
    """solve_mta.  ``return_mode=False``: lora_train_vlp.py:742-811 / ood.py:751-820
    (returns ``mode @ text * 100`` [1, C]);  ``return_mode=True``:
    slow_pace.py:1363-1433 / test.py (returns the mode feature [1, d]).
    image_features [V, d] unit rows (row 0 = centre view), text_features [d, C]."""
    logits = image_features @ text_features * 100
    lambda_y, lambda_q, max_iter, th = 0.2, 4, 5, 1e-6
    V = image_features.shape[0]
    dist = cdist(image_features, image_features)
    sorted_dist = jt_argsort_values(dist, dim=1)
    k = int(0.3 * (V - 1))
    sel = sorted_dist[:, 1:k + 1] ** 2
    bandwidth = torch.sqrt(0.5 * sel.mean(dim=1))
    sm = jt_softmax(logits, dim=1)
    affinity = sm @ sm.t()
    y = torch.ones(V, dtype=image_features.dtype) / V
    mode = image_features[0]
    n_y, n_m = [], []
    for _ in range(max_iter):
        density = gaussian_kernel(mode, bandwidth, image_features)
        i = 0
        while True:
            i += 1
            old_y = y
            y = jt_softmax(1 / lambda_y * (density + lambda_q * (affinity * y.unsqueeze(0)).sum(dim=1)), dim=-1)
            if torch.sqrt(((old_y - y) ** 2).sum()) < th or i >= max_iter:
                break
        n_y.append(i)
        i = 0
        while True:
            i += 1
            old_mode = mode
            density = gaussian_kernel(mode, bandwidth, image_features)
            wd = density * y
            mode = (wd.unsqueeze(1) * image_features).sum(dim=0) / wd.sum()
            mode = mode / torch.sqrt((mode * mode).sum())
            if torch.sqrt(((old_mode - mode) ** 2).sum()) < th or i >= max_iter:
                break
        n_m.append(i)
    out = mode.unsqueeze(0) if return_mode else mode.unsqueeze(0) @ text_features * 100
    if return_trace:
        return out, {"y": y, "mode": mode, "bandwidth": bandwidth, "n_y": n_y, "n_m": n_m}
    return out


def fuse_top5(cos: Tensor, cos1: Tensor, cos3: Tensor, head_logits: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """test.py:1729-1742: cos2=(cos+cos1)/2, cos4=(cos2+cos3)/2, cos5=cos4+0.5*head;
    the written top-5 uses cosine_similarity1 only (:1738)."""
    cos2 = (cos + cos1) / 2
    cos4 = (cos2 + cos3) / 2
    out = {"cos2": cos2, "cos4": cos4, "top5": jt_topk(cos1, 5)}
    if head_logits is not None:
        out["cos5"] = cos4 + 0.5 * head_logits
    return out


# --------------------------------------------------------------------------
# One stage-1 training step  (lora_train_vlp.py:956-1002)
# --------------------------------------------------------------------------


def train_step_loss(sd: Dict[str, Tensor], images: Tensor, captions: Tensor, target: Tensor,
                    text_lora: Optional[Dict[int, dict]], vis_lora: Optional[Dict[int, dict]], scaling: float,
                    class_index: Optional[Sequence[int]] = None, logit_scale: float = 100.0,
                    text_drops=None, vis_drops=None, ctx: Optional[Tensor] = None,
                    text_chunk: int = 32) -> Tuple[Tensor, Tensor]:
    """Loss of one step of run_lora's loop: text encoded in chunks of 32 WITH grad
    (:907-919,976), per class normalise/mean/normalise (:978-990), image encode
    (:992), 100*cos logits (:993-995), mean CE (:997).  With ``ctx`` the text side
    uses learnable prompt tokens (slow_pace.py:1626-1629).  Returns (loss, logits)."""
    C = captions.shape[0]
    embs = []
    for s in range(0, C, text_chunk):
        tk = captions[s:s + text_chunk]
        td = None
        if text_drops is not None:
            td = {blk: {p: m[:, s:s + text_chunk] for p, m in d.items()} for blk, d in text_drops.items()}
        if ctx is None:
            embs.append(encode_text(sd, tk, text_lora, scaling, drops=td))
        else:
            pe = build_prompts(ctx, sd["token_embedding.weight"], tk)
            embs.append(encode_text(sd, tk, text_lora, scaling, embeds=pe, drops=td))
    emb = torch.cat(embs, dim=0)
    if class_index is None:
        class_index = list(range(C))
    n_classes = max(class_index) + 1
    txt = class_text_features(emb, class_index, n_classes)
    img = encode_image(sd, images, vis_lora, scaling, drops=vis_drops)
    logits = train_logits(img, txt, logit_scale)
    return jt_cross_entropy(logits, target), logits


# --------------------------------------------------------------------------
# Tokenizer restatement  (jclip/simple_tokenizer.py, jclip/clip.py:190-214)
# --------------------------------------------------------------------------


def _bytes_to_unicode() -> Dict[int, str]:
    """simple_tokenizer.py:16-41 (GPT-2 printable byte alphabet)."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    chars = keep[:]
    extra = 0
    for b in range(256):
        if b not in keep:
            keep.append(b)
            chars.append(256 + extra)
            extra += 1
    return {b: chr(c) for b, c in zip(keep, chars)}


class OracleTokenizer:
    """SimpleTokenizer (simple_tokenizer.py:67-149) restated.  ``ftfy.fix_text``
    (:55) is treated as the identity (true for the ASCII class names / templates
    of this task; ftfy is not installed)."""

    def __init__(self, bpe_path: str):
        import gzip
        raw = open(bpe_path, "rb").read()
        if raw[:2] == b"\x1f\x8b":  # the shipped *.txt is in fact the gzip
            raw = gzip.decompress(raw)
        lines = raw.decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in lines[1:49152 - 256 - 2 + 1]]
        alphabet = list(_bytes_to_unicode().values())
        vocab = alphabet + [c + "</w>" for c in alphabet] + ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {t: i for i, t in enumerate(vocab)}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.byte_enc = _bytes_to_unicode()
        import regex
        self.pat = regex.compile(
            r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
            regex.IGNORECASE)

    def _bpe(self, token: str) -> List[str]:
        if token in ("<|startoftext|>", "<|endoftext|>"):  # pre-seeded cache entries, simple_tokenizer.py:83-86
            return [token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for a, b in zip(word[:-1], word[1:]):
                r = self.ranks.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and (word[i], word[i + 1]) == best:
                    merged.append(word[i] + word[i + 1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        return word

    def encode(self, text: str) -> List[int]:
        import html
        import re
        text = html.unescape(html.unescape(text)).strip()
        text = re.sub(r"\s+", " ", text).strip().lower()
        ids: List[int] = []
        for tok in self.pat.findall(text):
            tok = "".join(self.byte_enc[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._bpe(tok))
        return ids

    def tokenize(self, texts, context_length: int = 77, truncate: bool = False) -> Tensor:
        """jclip/clip.py:190-214."""
        if isinstance(texts, str):
            texts = [texts]
        sot, eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        out = torch.zeros(len(texts), context_length, dtype=torch.int64)
        for i, t in enumerate(texts):
            toks = [sot] + self.encode(t) + [eot]
            if len(toks) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                toks = toks[:context_length]
                toks[-1] = eot
            out[i, :len(toks)] = torch.tensor(toks)
        return out
