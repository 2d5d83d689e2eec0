"""Tensor-level wrappers over the C ABI (include/clipfs.h).  PyTorch is used only to own device
memory and streams; every function here enqueues hand-written HIP kernels on the current stream.
All tensors must be contiguous fp32 CUDA(ROCm) tensors unless stated otherwise."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import GemmArgs, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "clipfs ops need device tensors (there is no CPU path)"
    return t.data_ptr()


def _f32(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


_COUNTERS = {}


def _gemm_counters(device, n: int) -> torch.Tensor:
    """Zeroed stream-K arrival counters for a stand-alone GEMM call: cached per (device, stream) -- launches on one
    stream are ordered and every launch leaves the counters zero."""
    key = (device.index, _stream())
    buf = _COUNTERS.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.zeros(max(n, 4096), device=device, dtype=torch.int32)
        _COUNTERS[key] = buf
    return buf


def gemm_nt(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, *, bias=None, residual=None,
            act: int = 0, aux_out=None, aux_in=None, alpha: float = 1.0, lora_t=None, lora_b=None,
            lora_seg_width: int = 0, lora_scale: float = 0.0, split_k: bool = True, b_planes=None,
            a16: Optional[torch.Tensor] = None, out16: Optional[torch.Tensor] = None, only16: bool = False,
            aux_f16: bool = False) -> torch.Tensor:
    """out = epi(alpha * a @ b.T); a [M,K], b [N,K].  ``a16`` (f16 [M,K]) with f16 ``b_planes`` selects the
    f16 x f16 kernel (``a`` may then be None); ``out16`` (f16 [M,N]) receives an f16 copy of the result."""
    if a is not None:
        _f32(a)
    _f32(b)
    M, K = (a16 if a is None else a).shape
    N = b.shape[0]
    assert b.shape[1] == K
    if out is None and not only16:
        out = torch.empty(M, N, device=b.device, dtype=torch.float32)
    g = _lib.new_gemm_args()
    g.A, g.B, g.C = _p(a), _p(b), (None if only16 else _p(out))
    if a16 is not None:
        assert a16.dtype == torch.float16 and a16.is_contiguous() and tuple(a16.shape) == (M, K)
        g.A_f16 = _p(a16)
    if out16 is not None:
        assert out16.dtype == torch.float16 and out16.is_contiguous() and tuple(out16.shape) == (M, N)
        g.C_f16 = _p(out16)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = K, K, N
    g.alpha = alpha
    g.bias = _p(bias)
    g.residual = _p(residual)
    g.ldres = N
    g.act = act
    g.aux_out = _p(aux_out)
    g.aux_in = _p(aux_in)
    g.aux_f16 = int(aux_f16)  # f16 x f16 kernel: aux_out / aux_in are f16 tensors (fp16 storage of the pre-activation)
    if lora_t is not None:
        r = lora_b.shape[1]
        g.lora_t, g.lora_b = _p(_f32(lora_t)), _p(_f32(lora_b))
        g.lora_r = r
        g.lora_nseg = lora_t.shape[1] // r
        g.lora_seg_width = lora_seg_width or N
        g.lora_scale = lora_scale
    g.B_planes = _p(b_planes)
    if b_planes is not None:
        g.b_format = 2 if b_planes.dtype == torch.float16 else 1
    lib = _lib.load()
    ws = None
    nws = lib.clipfs_gemm_workspace_floats(M, N, K) if split_k else 0
    if nws:
        ws = torch.empty(nws, device=b.device, dtype=torch.float32)
        g.workspace, g.workspace_floats = _p(ws), nws
        ncnt = lib.clipfs_gemm_counter_ints(M, N, K)
        if ncnt and a is not None and b_planes is None:
            cnt = _gemm_counters(b.device, ncnt)
            g.counters, g.counters_ints = _p(cnt), cnt.numel()
    check(lib.clipfs_gemm_nt(C.byref(g), _stream()), "gemm_nt")
    return out16 if only16 else out


def split_bf16(w: torch.Tensor) -> torch.Tensor:
    """hi/lo bf16 planes of a frozen fp32 weight: int16 tensor [2, *w.shape] (bit patterns)."""
    _f32(w)
    planes = torch.empty((2,) + tuple(w.shape), device=w.device, dtype=torch.int16)
    check(_lib.load().clipfs_split_bf16(_p(w), _p(planes), w.numel(), _stream()), "split_bf16")
    return planes


def to_f16(w: torch.Tensor) -> torch.Tensor:
    """f16 plane of a frozen fp32 weight (fp16 MFMA mode)."""
    _f32(w)
    out = torch.empty(w.shape, device=w.device, dtype=torch.float16)
    check(_lib.load().clipfs_convert_f16(_p(w), _p(out), w.numel(), _stream()), "convert_f16")
    return out


def patch_embed(images: torch.Tensor, conv_w: torch.Tensor, pos: torch.Tensor, x: torch.Tensor, tokens: int) -> None:
    """x[b, 1+p, :] = conv(images)[b, :, p] + pos[1+p]  (im2col-free GEMM on the NCHW batch)."""
    _f32(images), _f32(conv_w), _f32(pos), _f32(x)
    B, ch, R, _ = images.shape
    width, _, ps, _ = conv_w.shape
    assert ch == 3
    P = (R // ps) ** 2
    g = _lib.new_gemm_args()
    g.A, g.B, g.C = _p(images), _p(conv_w), _p(x)
    g.M, g.N, g.K = B * P, width, 3 * ps * ps
    g.lda, g.ldb, g.ldc = 0, 3 * ps * ps, width
    g.alpha = 1.0
    g.residual = _p(pos)
    g.ldres = width
    g.a_mode = 1
    g.img_res, g.patch, g.out_tokens = R, ps, tokens
    check(_lib.load().clipfs_gemm_nt(C.byref(g), _stream()), "patch_embed")


def layernorm_fwd(x, gamma, beta, *, ldx=None, rows=None, save_stats=False, eps=1e-5):
    width = gamma.numel()
    if rows is None:
        rows = x.numel() // width
    ldx = ldx or width
    y = torch.empty(rows, width, device=x.device, dtype=torch.float32)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(_lib.load().clipfs_layernorm_fwd(_p(x), ldx, _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, width,
                                           eps, _stream()), "layernorm_fwd")
    return (y, mean, rstd) if save_stats else y


def layernorm_fwd_lora(x, gamma, beta, A, r, nseg=3, *, seg_mask=None, p=0.0, seed=0, stream_base=0, row0=0, eps=1e-5,
                       keep_bits=None):
    """LayerNorm + adapter down-projection in one pass: returns (y, t, mean, rstd)."""
    width = gamma.numel()
    rows = x.numel() // width
    lib = _lib.load()
    if not lib.clipfs_layernorm_fwd_lora_ok(width, r, nseg):
        raise ValueError(f"layernorm_fwd_lora: width {width} r {r} nseg {nseg} is not covered")
    y = torch.empty(rows, width, device=x.device, dtype=torch.float32)
    t = torch.empty(rows, nseg * r, device=x.device, dtype=torch.float32)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    if seg_mask is None:
        seg_mask = (1 << nseg) - 1
    check(lib.clipfs_layernorm_fwd_lora(_p(_f32(x)), width, _p(gamma), _p(beta), _p(y), None, _p(mean), _p(rstd), rows, width,
                                        eps, _p(_f32(A)), _p(t), r, nseg, seg_mask, p, seed, stream_base, row0, _p(keep_bits),
                                        _stream()),
          "layernorm_fwd_lora")
    return y, t, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, *, ldx=None, dres=None, dx=None, lddx=None):
    width = gamma.numel()
    rows = dy.numel() // width
    ldx = ldx or width
    lddx = lddx or width
    if dx is None:
        dx = torch.empty(rows, width, device=dy.device, dtype=torch.float32)
    check(_lib.load().clipfs_layernorm_bwd(_p(dy), _p(x), ldx, _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), lddx,
                                           rows, width, _stream()), "layernorm_bwd")
    return dx


def attention_fwd(qkv, batch, seq, heads, causal, want_lse=False):
    out = torch.empty(batch * seq, heads * 64, device=qkv.device, dtype=torch.float32)
    lib = _lib.load()
    n = lib.clipfs_attention_lse_floats(batch, seq, heads) if want_lse else 0
    lse = torch.empty(n, device=qkv.device, dtype=torch.float32) if n else None
    check(lib.clipfs_attention_fwd(_p(_f32(qkv)), _p(out), _p(lse), batch, seq, heads, int(causal), _stream()),
          "attention_fwd")
    return (out, lse) if want_lse else out


def attention_f16_fwd(qkv, batch, seq, heads, causal=False):
    """fp16-mode MFMA attention (seq <= 288): returns (out, lse)."""
    out = torch.empty(batch * seq, heads * 64, device=qkv.device, dtype=torch.float32)
    lse = torch.empty(batch * heads * seq, device=qkv.device, dtype=torch.float32)
    f16 = qkv.dtype == torch.float16
    check(_lib.load().clipfs_attention_f16_fwd(_p(qkv if f16 else _f32(qkv)), int(f16), _p(out), None, _p(lse), batch, seq,
                                               heads, int(causal), _stream()), "attention_f16_fwd")
    return out, lse


def attention_f16_bwd(qkv, dout, out, lse, batch, seq, heads, causal=False):
    f16 = qkv.dtype == torch.float16
    dqkv = torch.empty(qkv.shape, device=qkv.device, dtype=torch.float32)
    work = torch.empty_like(lse)
    g16 = dout.dtype == torch.float16  # dO as its f16 image (what the fp16-storage tower hands over) or fp32
    assert dout.is_contiguous()
    check(_lib.load().clipfs_attention_f16_bwd(_p(qkv if f16 else _f32(qkv)), int(f16), _p(dout if g16 else _f32(dout)), int(g16),
                                               _p(out), _p(lse), _p(dqkv), None, _p(work), batch, seq, heads, int(causal), _stream()),
          "attention_f16_bwd")
    return dqkv


def attention_bwd(qkv, dout, batch, seq, heads, causal, out=None, lse=None):
    dqkv = torch.empty_like(qkv)
    work = torch.empty_like(lse) if lse is not None else None
    check(_lib.load().clipfs_attention_bwd(_p(_f32(qkv)), _p(_f32(dout)), _p(out), _p(lse), _p(dqkv), _p(work), batch,
                                           seq, heads, int(causal), _stream()), "attention_bwd")
    return dqkv


def lora_keep_bits(rows, width, device):
    """Buffer for the dropout masks an adapted projection records in its forward (uint16 per float4 of the input)."""
    return torch.zeros(rows, width // 4, device=device, dtype=torch.int16)


def lora_down(x, A, r, nseg, seg_mask=None, p=0.0, seed=0, stream_base=0, row0=0, keep_bits=None):
    rows, width = x.shape
    t = torch.empty(rows, nseg * r, device=x.device, dtype=torch.float32)
    if seg_mask is None:
        seg_mask = (1 << nseg) - 1
    check(_lib.load().clipfs_lora_down(_p(_f32(x)), _p(_f32(A)), _p(t), rows, width, r, nseg, seg_mask, p, seed,
                                       stream_base, row0, _p(keep_bits), _stream()), "lora_down")
    return t


def lora_bwd(dy, x, t, A, B, dA, dB, *, dx=None, scale, p=0.0, seed=0, stream_base=0, seg_mask=None, row0=0, keep_bits=None):
    rows, width = x.shape
    nseg = dy.shape[1] // width
    r = B.shape[1]
    if seg_mask is None:
        seg_mask = (1 << nseg) - 1
    lib = _lib.load()
    nwork = lib.clipfs_lora_bwd_work_floats(rows, width, r, nseg)
    work = torch.empty(nwork, device=x.device, dtype=torch.float32)
    dt = torch.empty(rows, nseg * r, device=x.device, dtype=torch.float32)
    if dy.dtype == torch.float16:  # fp16 storage mode: the f16 image of the incoming gradient (matrix-core shapes only)
        assert dy.is_contiguous()
        check(lib.clipfs_lora_bwd_f16dy(_p(dy), _p(_f32(x)), _p(_f32(t)), _p(_f32(A)), _p(_f32(B)), _p(dt), _p(dA),
                                        _p(dB), _p(dx), rows, width, width, r, nseg, seg_mask, scale, p, seed, stream_base,
                                        row0, _p(keep_bits), _p(work), _stream()), "lora_bwd_f16dy")
        return dt
    check(lib.clipfs_lora_bwd(_p(_f32(dy)), _p(_f32(x)), _p(_f32(t)), _p(_f32(A)), _p(_f32(B)), _p(dt), _p(dA),
                              _p(dB), _p(dx), rows, width, width, r, nseg, seg_mask, scale, p, seed, stream_base,
                              row0, _p(keep_bits), _p(work), _stream()), "lora_bwd")
    return dt


def vit_fill_special(x, class_emb, pos, vpt, batch, tokens, n_patch):
    n_vpt = 0 if vpt is None else vpt.shape[0]
    check(_lib.load().clipfs_vit_fill_special(_p(x), _p(class_emb), _p(pos), _p(vpt), batch, tokens, n_patch, n_vpt,
                                              class_emb.numel(), _stream()), "vit_fill_special")


def text_embed(ids, table, pos, ctx=None):
    n, seq = ids.shape
    width = table.shape[1]
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    x = torch.empty(n * seq, width, device=table.device, dtype=torch.float32)
    n_ctx = 0 if ctx is None else ctx.shape[0]
    check(_lib.load().clipfs_text_embed(_p(ids), _p(table), _p(pos), _p(ctx), n_ctx, _p(x), n, seq, width, _stream()),
          "text_embed")
    return x


def token_rows_grad(dx, dtok, n, seq, first):
    n_tok, width = dtok.shape
    check(_lib.load().clipfs_token_rows_grad(_p(dx), _p(dtok), n, seq, width, n_tok, first, _stream()),
          "token_rows_grad")


def matmul_small(a, b, M, N, K, sam, sak, sbk, sbn, alpha=1.0, out=None):
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    check(_lib.load().clipfs_matmul_small(_p(a), _p(b), _p(out), M, N, K, sam, sak, sbk, sbn, alpha, _stream()),
          "matmul_small")
    return out


def eot_index(ids: torch.Tensor) -> torch.Tensor:
    """idx[c] = position of the EOT token (largest id, first maximum) of caption c, int32 [n] (jclip/model.py:213-214)."""
    assert ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous()
    n, seq = ids.shape
    idx = torch.empty(n, device=ids.device, dtype=torch.int32)
    check(_lib.load().clipfs_eot_index(_p(ids), _p(idx), n, seq, _stream()), "eot_index")
    return idx


def gather_eot(x, ids):
    n, seq = ids.shape
    width = x.shape[-1]
    out = torch.empty(n, width, device=x.device, dtype=torch.float32)
    idx = torch.empty(n, device=x.device, dtype=torch.int32)
    check(_lib.load().clipfs_gather_eot(_p(x), _p(ids), _p(out), _p(idx), n, seq, width, _stream()), "gather_eot")
    return out, idx


def scatter_rows(dy, idx, seq, out=None):
    n, width = dy.shape
    dx = out if out is not None else torch.empty(n * seq, width, device=dy.device, dtype=torch.float32)
    check(_lib.load().clipfs_scatter_rows(_p(_f32(dy)), _p(idx), _p(dx), n, seq, width, _stream()), "scatter_rows")
    return dx


def l2norm_fwd(x, save_inv=False):
    rows, width = x.shape
    y = torch.empty_like(x)
    inv = torch.empty(rows, device=x.device, dtype=torch.float32) if save_inv else None
    check(_lib.load().clipfs_l2norm_fwd(_p(_f32(x)), _p(y), _p(inv), rows, width, _stream()), "l2norm_fwd")
    return (y, inv) if save_inv else y


def l2norm_bwd(dy, y, inv):
    rows, width = y.shape
    dx = torch.empty_like(y)
    check(_lib.load().clipfs_l2norm_bwd(_p(_f32(dy)), _p(y), _p(inv), _p(dx), rows, width, _stream()), "l2norm_bwd")
    return dx


def class_mean_fwd(emb, classes, templates, out=None):
    width = emb.shape[1]
    if out is None:
        out = torch.empty(classes, width, device=emb.device, dtype=torch.float32)
    assert out.is_contiguous() and tuple(out.shape) == (classes, width)
    check(_lib.load().clipfs_class_mean_fwd(_p(_f32(emb)), _p(out), classes, templates, width, _stream()),
          "class_mean_fwd")
    return out


def class_mean_bwd(emb, dout, classes, templates):
    demb = torch.empty_like(emb)
    check(_lib.load().clipfs_class_mean_bwd(_p(_f32(emb)), _p(_f32(dout)), _p(demb), classes, templates, emb.shape[1],
                                            _stream()), "class_mean_bwd")
    return demb


def cross_entropy(logits, target, want_grad=True, grad_scale=1.0):
    """returns (loss_sum [1], dlogits or None, correct [1] int32)"""
    rows, classes = logits.shape
    assert target.dtype == torch.int64
    dl = torch.empty_like(logits) if want_grad else None
    loss_rows = torch.empty(2 * rows, device=logits.device, dtype=torch.float32)
    loss_sum = torch.empty(1, device=logits.device, dtype=torch.float32)
    correct = torch.empty(1, device=logits.device, dtype=torch.int32)
    check(_lib.load().clipfs_cross_entropy(_p(_f32(logits)), _p(target), _p(dl), _p(loss_rows), _p(loss_sum),
                                           _p(correct), rows, classes, grad_scale, _stream()), "cross_entropy")
    return loss_sum, dl, correct


def topk(logits, k):
    rows, classes = logits.shape
    labels = torch.empty(rows, k, device=logits.device, dtype=torch.int32)
    check(_lib.load().clipfs_topk(_p(_f32(logits)), _p(labels), rows, classes, k, _stream()), "topk")
    return labels


def channel_affine(x, scale1, bias1):
    y = torch.empty_like(x)
    check(_lib.load().clipfs_channel_affine(_p(_f32(x)), _p(scale1), _p(bias1), _p(y), x.shape[0], x.shape[1],
                                            _stream()), "channel_affine")
    return y


def logit_normalize(z):
    out = torch.empty_like(z)
    work = torch.empty(2, device=z.device, dtype=torch.float32)
    check(_lib.load().clipfs_logit_normalize(_p(_f32(z)), _p(out), _p(work), z.shape[0], z.shape[1], _stream()),
          "logit_normalize")
    return out


def logit_normalize_bwd(z, dzn):
    dz = torch.empty_like(z)
    check(_lib.load().clipfs_logit_normalize_bwd(_p(_f32(z)), _p(_f32(dzn)), _p(dz), z.shape[0], z.shape[1], _stream()),
          "logit_normalize_bwd")
    return dz


def colsum(x, y=None):
    out = torch.empty(x.shape[1], device=x.device, dtype=torch.float32)
    check(_lib.load().clipfs_colsum(_p(_f32(x)), _p(y), _p(out), x.shape[0], x.shape[1], _stream()), "colsum")
    return out


def adamw(p, g, m, v, step, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
    check(_lib.load().clipfs_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), step, lr, betas[0], betas[1], eps,
                                   weight_decay, grad_scale, _stream()), "adamw")


def mta(feats, text, want_mode=True, want_logits=True):
    """feats [n_img, V, d] unit rows, text [C, d] unit rows -> (mode [n_img,d], logits [n_img,C])"""
    n_img, V, d = feats.shape
    Cn = text.shape[0]
    lib = _lib.load()
    work = torch.empty(lib.clipfs_mta_work_floats(n_img, V, d, Cn), device=feats.device, dtype=torch.float32)
    mode = torch.empty(n_img, d, device=feats.device, dtype=torch.float32) if want_mode else None
    logits = torch.empty(n_img, Cn, device=feats.device, dtype=torch.float32) if want_logits else None
    check(lib.clipfs_mta(_p(_f32(feats)), _p(_f32(text)), _p(mode), _p(logits), _p(work), n_img, V, d, Cn, _stream()),
          "mta")
    return mode, logits


def l1_loss(a: torch.Tensor, b: torch.Tensor, want_grad: bool = False, grad_scale: float = 1.0):
    """mean |a - b| (slow_pace.py:1654-1655); optionally d/da scaled by grad_scale."""
    a, b = _f32(a).contiguous(), _f32(b).contiguous()
    assert a.shape == b.shape
    loss = torch.empty(1, device=a.device, dtype=torch.float32)
    da = torch.empty_like(a) if want_grad else None
    check(_lib.load().clipfs_l1_loss(_p(a), _p(b), a.numel(), _p(loss), _p(da), grad_scale, _stream()), "l1_loss")
    return (loss, da) if want_grad else loss


def kl_logits(logits: torch.Tensor, target_logits: torch.Tensor, want_grad: bool = False, grad_scale: float = 1.0):
    """per-row KL(softmax(target) || softmax(logits)) (slow_pace.py:1656-1658 with kl_div :1170-1177)."""
    x, t = _f32(logits).contiguous(), _f32(target_logits).contiguous()
    assert x.shape == t.shape and x.dim() == 2
    rows, cols = x.shape
    loss_rows = torch.empty(rows, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x) if want_grad else None
    check(_lib.load().clipfs_kl_logits(_p(x), _p(t), _p(loss_rows), _p(dx), rows, cols, grad_scale, _stream()), "kl_logits")
    return (loss_rows, dx) if want_grad else loss_rows
