"""Adapter API of the reference's ``lora_train_vlp.py`` on the HIP engine: same names, argument
meaning and error behaviour, different machinery.

    apply_lora / get_lora_parameters / mark_only_lora_as_trainable / lora_state_dict /
    save_lora / load_lora        lora_train_vlp.py:122-179,516-635
    LoRALayer / LinearLoRA / PlainMultiheadAttentionLoRA      :185-306,377-513
    cls_acc / clip_classifier / encode_text_in_batches / solve_mta   :638-666,742-811,907-919
    LoRATrainer.step             the body of run_lora's loop, :956-1002 (+ AdamW :946)

What changes underneath: the q/k/v adapters of one attention block live in ONE stacked pair
(A [3r, d], B [3d, r]; ``q_proj.w_lora_A`` etc. are views), the low-rank update is applied in rank-r form
inside the QKV GEMM epilogue instead of materialising B@A (:218-221,302), and all trainable tensors of a
model are re-homed into one flat fp32 buffer so the data-parallel gradient exchange is a single
all-reduce and the optimiser a single fused launch."""
from __future__ import annotations

import math
import os
import pickle
from typing import Dict, Optional, Sequence

import numpy as np
import torch
from torch import nn

from clipfs import ops, safe_pkl
from jclip import clip

INDEX_POSITIONS_TEXT = {
    'top1': [11], 'top2': [10, 11], 'top3': [9, 10, 11], 'bottom': [0, 1, 2, 3], 'mid': [4, 5, 6, 7],
    'up': [8, 9, 10, 11], 'half-up': [6, 7, 8, 9, 10, 11], 'half-bottom': [0, 1, 2, 3, 4, 5],
    'all': [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]}

INDEX_POSITIONS_VISION = {
    'ViT-B/16': {'top': [11], 'top3': [9, 10, 11], 'bottom': [0, 1, 2, 3], 'mid': [4, 5, 6, 7], 'up': [8, 9, 10, 11],
                 'half-up': [6, 7, 8, 9, 10, 11], 'half-bottom': [0, 1, 2, 3, 4, 5],
                 'all': [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]},
    'ViT-B/32': {'bottom': [0, 1, 2, 3], 'mid': [4, 5, 6, 7], 'up': [8, 9, 10, 11], 'half-up': [6, 7, 8, 9, 10, 11],
                 'half-bottom': [0, 1, 2, 3, 4, 5], 'all': [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]},
    # reference quirk kept: 'all' stops at block 20 of 24 (lora_train_vlp.py:62)
    'ViT-L/14': {'bottom': [0, 1, 2, 3], 'mid': [4, 5, 6, 7], 'up': [8, 9, 10, 11], 'half-up': [6, 7, 8, 9, 10, 11],
                 'half-bottom': [0, 1, 2, 3, 4, 5], 'all': list(range(21))},
}

_PROJ_BIT = {'q': 1, 'k': 2, 'v': 4, 'o': 8}
_PROJ_NAME = {'q': 'q_proj', 'k': 'k_proj', 'v': 'v_proj', 'o': 'proj'}


# ----------------------------------------------------------------------------------------------
# parameter filters (lora_train_vlp.py:122-179)
# ----------------------------------------------------------------------------------------------

def get_lora_parameters(model, bias='none'):
    params = []
    named = dict(model.named_parameters())
    for name, param in named.items():
        if bias == 'none':
            if 'lora_' in name:
                params.append(param)
        elif bias == 'all':
            if 'lora_' in name or 'bias' in name:
                params.append(param)
        elif bias == 'lora_only':
            if 'lora_' in name:
                params.append(param)
                bias_name = name.split('lora_')[0] + 'bias'
                if bias_name in named:
                    params.append(named[bias_name])
        else:
            raise NotImplementedError
    return params


def mark_only_lora_as_trainable(model, bias: str = 'none') -> None:
    """lora_train_vlp.py:143-160: freeze everything whose name has no ``lora_``; ``bias='all'`` re-enables every
    parameter named ``*bias*``, ``'lora_only'`` the bias of each LoRA-wrapped linear, anything else raises
    NotImplementedError.  The flags are set exactly as the reference sets them (so ``get_lora_parameters`` /
    optimiser parameter lists come out the same).  NOTE: the reference's own training loop never uses a mode other than
    'none' (its optimiser is built from ``get_lora_parameters(model)``, lora_train_vlp.py:946); the fused HIP backward
    computes adapter / prompt gradients only, so a bias enabled here receives no gradient from ``LoRATrainer`` -- which is
    also what the reference does with it, since ``p.requires_grad_ = True`` (:151,:157) assigns an attribute instead of
    calling the method."""
    for n, p in model.named_parameters():
        if 'lora_' not in n:
            p.requires_grad_(False)
    if bias == 'none':
        return
    if bias == 'all':
        for n, p in model.named_parameters():
            if 'bias' in n:
                p.requires_grad_(True)
    elif bias == 'lora_only':
        for m in model.modules():
            if isinstance(m, LoRALayer) and getattr(m, 'bias', None) is not None:
                m.bias.requires_grad_(True)
    else:
        raise NotImplementedError
    import warnings
    warnings.warn(f"mark_only_lora_as_trainable(bias={bias!r}): the flags are set as the reference sets them, but the fused "
                  "backward (LoRATrainer) produces adapter / prompt gradients only -- these biases will NOT be trained")


def lora_state_dict(model, bias: str = 'none'):
    sd = model.state_dict()
    if bias == 'none':
        return {k: sd[k] for k in sd if 'lora_' in k}
    if bias == 'all':
        return {k: sd[k] for k in sd if 'lora_' in k or 'bias' in k}
    if bias == 'lora_only':
        out = {}
        for k in sd:
            if 'lora_' in k:
                out[k] = sd[k]
                bn = k.split('lora_')[0] + 'bias'
                if bn in sd:
                    out[bn] = sd[bn]
        return out
    raise NotImplementedError


# ----------------------------------------------------------------------------------------------
# LoRA layers
# ----------------------------------------------------------------------------------------------

class LoRALayer:
    """lora_train_vlp.py:185-245 (state only; merging is never needed: the engine always applies the
    additive form, which is the branch the reference executes)."""

    def __init__(self, r: int, lora_alpha: int, fan_in_fan_out: bool = False, dropout_rate: float = 0):
        self.r = r
        self.lora_alpha = lora_alpha
        self.dropout_rate = dropout_rate
        if self.r > 0:
            self.scaling = self.lora_alpha / math.sqrt(self.r)  # :197 -- alpha / sqrt(r), not alpha / r
        self.merged = False
        self.fan_in_fan_out = fan_in_fan_out
        self.params_with_lora = {}


class LinearLoRA(nn.Module, LoRALayer):
    """One adapted projection (lora_train_vlp.py:248-306).  ``weight`` / ``bias`` / ``w_lora_A`` /
    ``w_lora_B`` are views into the owning attention block's stacked tensors."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], a_view: torch.Tensor, b_view: torch.Tensor,
                 r: int, lora_alpha: int, dropout_rate: float):
        nn.Module.__init__(self)
        LoRALayer.__init__(self, r=r, lora_alpha=lora_alpha, dropout_rate=dropout_rate)
        self.weight = nn.Parameter(weight, requires_grad=False)
        self.bias = nn.Parameter(bias, requires_grad=False) if bias is not None else None
        self.out_features, self.in_features = weight.shape
        self.params_with_lora = {'weight': 'w'}
        self.w_lora_A = nn.Parameter(a_view)
        self.w_lora_B = nn.Parameter(b_view)

    def merge_BA(self, param_name: str = 'weight') -> torch.Tensor:
        """B @ A (:218-221); host-side helper for inspection / merged-weight checks."""
        return (self.w_lora_B.data @ self.w_lora_A.data).reshape(self.weight.shape)


class _FrozenLinear(nn.Module):
    def __init__(self, weight, bias):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=False)
        self.bias = nn.Parameter(bias, requires_grad=False) if bias is not None else None
        self.out_features, self.in_features = weight.shape


class PlainMultiheadAttentionLoRA(nn.Module, LoRALayer):
    """lora_train_vlp.py:377-513.  Splits the packed in-projection into q/k/v views (rows [0:d], [d:2d],
    [2d:3d], :395-409) WITHOUT copying -- the stacked tensor stays the GEMM operand -- and attaches
    LoRA pairs to the projections named in ``enable_lora``."""

    is_lora_mha = True

    def __init__(self, existing_mha, enable_lora: Sequence[str] = ('q', 'k', 'v', 'o'), r: int = 0, lora_alpha: int = 1,
                 dropout_rate: float = 0., seed: Optional[int] = None, **kwargs):
        nn.Module.__init__(self)
        LoRALayer.__init__(self, r=r, lora_alpha=lora_alpha, dropout_rate=dropout_rate)
        self.dropout = 0
        self.embed_dim = d = existing_mha.embed_dim
        self.kdim, self.vdim = existing_mha.kdim, existing_mha.vdim
        self._qkv_same_embed_dim = existing_mha._qkv_same_embed_dim
        self.num_heads = existing_mha.num_heads
        self.batch_first = existing_mha.batch_first
        self.head_dim = existing_mha.head_dim
        dev = existing_mha.in_proj_weight.device
        # stacked base weights (shared storage with the module being replaced)
        self.qkv_weight = existing_mha.in_proj_weight.data
        self.qkv_bias = existing_mha.in_proj_bias.data
        self._o_w = existing_mha.out_proj.weight.data
        self._o_b = existing_mha.out_proj.bias.data
        self.enable_lora = list(enable_lora)
        self.lora_mask = 0
        for item in self.enable_lora:
            if item not in _PROJ_BIT:
                raise ValueError(f"unknown projection {item!r} (expected q, k, v, o)")
            self.lora_mask |= _PROJ_BIT[item]
        if r <= 0:
            self.lora_mask = 0
        # stacked adapter storage; A ~ kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(in)), B = 0  (:209-213)
        gen = torch.Generator(device="cpu")
        gen.manual_seed(torch.initial_seed() if seed is None else seed)
        rr = max(r, 1)
        bound = 1.0 / math.sqrt(d)

        def init_a(rows):
            return ((torch.rand(rows, d, generator=gen) * 2 - 1) * bound).to(dev)

        self.lora_A_qkv = torch.zeros(3 * rr, d, device=dev)
        self.lora_B_qkv = torch.zeros(3 * d, rr, device=dev)
        self.grad_A_qkv = torch.zeros_like(self.lora_A_qkv)
        self.grad_B_qkv = torch.zeros_like(self.lora_B_qkv)
        self.lora_A_o = torch.zeros(rr, d, device=dev)
        self.lora_B_o = torch.zeros(d, rr, device=dev)
        self.grad_A_o = torch.zeros_like(self.lora_A_o)
        self.grad_B_o = torch.zeros_like(self.lora_B_o)
        for s, item in enumerate(('q', 'k', 'v')):
            if self.lora_mask & _PROJ_BIT[item]:
                self.lora_A_qkv[s * rr:(s + 1) * rr] = init_a(rr)
        if self.lora_mask & 8:
            self.lora_A_o[:] = init_a(rr)
        self._bind_views()

    def _bind_views(self):
        d, r = self.embed_dim, max(self.r, 1)
        mods = {}
        for s, (item, name) in enumerate((('q', 'q_proj'), ('k', 'k_proj'), ('v', 'v_proj'))):
            w, b = self.qkv_weight[s * d:(s + 1) * d], self.qkv_bias[s * d:(s + 1) * d]
            if self.lora_mask & _PROJ_BIT[item]:
                mods[name] = LinearLoRA(w, b, self.lora_A_qkv[s * r:(s + 1) * r], self.lora_B_qkv[s * d:(s + 1) * d],
                                        self.r, self.lora_alpha, self.dropout_rate)
            else:
                mods[name] = _FrozenLinear(w, b)
        ow, ob = self._o_w, self._o_b
        if self.lora_mask & 8:
            mods['proj'] = LinearLoRA(ow, ob, self.lora_A_o, self.lora_B_o, self.r, self.lora_alpha, self.dropout_rate)
        else:
            mods['proj'] = _FrozenLinear(ow, ob)
        for k, m in mods.items():
            old = getattr(self, k, None)
            if isinstance(old, LinearLoRA) and isinstance(m, LinearLoRA):
                m.w_lora_A.requires_grad_(old.w_lora_A.requires_grad)
                m.w_lora_B.requires_grad_(old.w_lora_B.requires_grad)
            setattr(self, k, m)

    # -- engine hooks -------------------------------------------------------------------------------
    def stacked(self):
        """[(name, param tensor, grad tensor)] of the stacked trainables of this block."""
        out = []
        if self.lora_mask & 7:
            out += [("lora_A_qkv", self.lora_A_qkv, self.grad_A_qkv), ("lora_B_qkv", self.lora_B_qkv, self.grad_B_qkv)]
        if self.lora_mask & 8:
            out += [("lora_A_o", self.lora_A_o, self.grad_A_o), ("lora_B_o", self.lora_B_o, self.grad_B_o)]
        return out

    def rebind(self, name: str, param: torch.Tensor, grad: torch.Tensor):
        """Move one stacked tensor into externally owned storage (flat buffer) and refresh the views."""
        setattr(self, name, param)
        setattr(self, "grad_" + name[len("lora_"):], grad)
        self._bind_views()

    def trainable_pairs(self):
        d, r = self.embed_dim, max(self.r, 1)
        out = []
        for s, (item, name) in enumerate((('q', 'q_proj'), ('k', 'k_proj'), ('v', 'v_proj'))):
            if self.lora_mask & _PROJ_BIT[item]:
                m = getattr(self, name)
                out.append((m.w_lora_A, self.grad_A_qkv[s * r:(s + 1) * r]))
                out.append((m.w_lora_B, self.grad_B_qkv[s * d:(s + 1) * d]))
        if self.lora_mask & 8:
            out.append((self.proj.w_lora_A, self.grad_A_o))
            out.append((self.proj.w_lora_B, self.grad_B_o))
        return out

    @torch.no_grad()
    def forward(self, query, key=None, value=None, need_weights=False, attn_mask=None, **_):
        """Direct call of one adapted attention block (lora_train_vlp.py:431-513): [L, N, d] in, ([L, N, d], None)
        out, causal iff an ``attn_mask`` is given -- the same kernels the fused tower sequences (rank-r LoRA down
        projection with the Philox dropout of train mode, QKV GEMM with the LoRA up-projection in its epilogue,
        attention, output projection).  Inference only, like ``jclip.mha.MultiheadAttention.forward``: training
        differentiates through ``encode_image`` / ``encode_text`` (the tower), not through single blocks."""
        if need_weights:
            raise NotImplementedError("attention weights are never materialised (need_weights=False only)")
        L, N, d = query.shape
        if attn_mask is not None:
            # the kernels implement exactly one mask: the causal -inf-above-the-diagonal mask of the text tower
            # (jclip/model.py:189-193); anything else would be applied silently wrong
            am = torch.as_tensor(attn_mask)
            causal = torch.full((L, L), float("-inf"), device=am.device, dtype=torch.float32).triu_(1)
            if tuple(am.shape) != (L, L) or not torch.equal(am.float(), causal):
                raise NotImplementedError("attn_mask must be the causal mask (-inf above the diagonal, 0 elsewhere)")
        r = self.r
        x = query.permute(1, 0, 2).reshape(N * L, d).contiguous().float()
        p = float(self.dropout_rate) if self.training else 0.0
        seed = 0
        if p > 0:
            self._direct_calls = getattr(self, "_direct_calls", 0) + 1
            seed = (0x9E3779B97F4A7C15 * self._direct_calls + 0x5EED) & 0xFFFFFFFFFFFFFFFF or 1
        qkv_mask = self.lora_mask & 7
        if qkv_mask and r > 0:
            t = ops.lora_down(x, self.lora_A_qkv, r, 3, seg_mask=qkv_mask, p=p, seed=seed, stream_base=0)
            qkv = ops.gemm_nt(x, self.qkv_weight, bias=self.qkv_bias, lora_t=t, lora_b=self.lora_B_qkv, lora_seg_width=d,
                              lora_scale=float(self.scaling))
        else:
            qkv = ops.gemm_nt(x, self.qkv_weight, bias=self.qkv_bias)
        o = ops.attention_fwd(qkv, N, L, self.num_heads, attn_mask is not None)
        if (self.lora_mask & 8) and r > 0:
            t = ops.lora_down(o, self.lora_A_o, r, 1, p=p, seed=seed, stream_base=3)
            y = ops.gemm_nt(o, self._o_w, bias=self._o_b, lora_t=t, lora_b=self.lora_B_o, lora_seg_width=d,
                            lora_scale=float(self.scaling))
        else:
            y = ops.gemm_nt(o, self._o_w, bias=self._o_b)
        return y.reshape(N, L, d).permute(1, 0, 2).contiguous(), None

    execute = forward


def apply_lora(args, clip_model):
    """lora_train_vlp.py:516-548: text blocks first, then vision blocks; a block is adapted when its
    ``attn`` is (still) a ``MultiheadAttention``."""
    list_lora_layers = []

    def adapt(blocks, indices):
        for i, block in enumerate(blocks):
            if i in indices:
                sub = block.attn
                if sub.__class__.__name__ == 'MultiheadAttention':
                    new = PlainMultiheadAttentionLoRA(sub, enable_lora=args.params, r=args.r, lora_alpha=args.alpha,
                                                      dropout_rate=args.dropout_rate)
                    block.attn = new
                    list_lora_layers.append(new)

    if args.encoder == 'text' or args.encoder == 'both':
        adapt(clip_model.transformer.resblocks, INDEX_POSITIONS_TEXT[args.position])
    if args.encoder == 'vision' or args.encoder == 'both':
        adapt(clip_model.visual.transformer.resblocks, INDEX_POSITIONS_VISION[args.backbone][args.position])
    clip_model.invalidate_engine()
    return list_lora_layers


# ----------------------------------------------------------------------------------------------
# checkpoint schema (lora_train_vlp.py:551-635)
# ----------------------------------------------------------------------------------------------

def _layer_weights(args, layer) -> dict:
    out = {}
    for p in ('q', 'k', 'v', 'o'):
        if p in args.params:
            m = getattr(layer, _PROJ_NAME[p])
            out[_PROJ_NAME[p]] = {'w_lora_A': m.w_lora_A.detach().cpu().numpy().copy(),
                                  'w_lora_B': m.w_lora_B.detach().cpu().numpy().copy()}
    return out


def save_lora(args, epoch, list_lora_layers, save_path: str = 'lora_weights1/lora_weights.pkl'):
    """Same schema and default path as the reference (:551-593); a plain pickle of numpy arrays, which is
    what ``jt.save`` writes, so either side can read the other's file."""
    weights = {f'layer_{i}': _layer_weights(args, layer) for i, layer in enumerate(list_lora_layers)}
    metadata = {'r': args.r, 'alpha': args.alpha, 'encoder': args.encoder, 'params': args.params,
                'position': args.position}
    os.makedirs(os.path.dirname(save_path) or '.', exist_ok=True)
    with open(save_path, 'wb') as f:
        pickle.dump({'weights': weights, 'metadata': metadata}, f, protocol=4)
    print(f'LoRA weights saved to {save_path}')


def load_lora(args, list_lora_layers, load_path):
    """:596-635 -- FileNotFoundError when missing, ValueError on any metadata mismatch.  The file is read
    with the inert reader (clipfs.safe_pkl): nothing in it is executed."""
    if not os.path.exists(load_path):
        raise FileNotFoundError(f'File {load_path} does not exist.')
    loaded = safe_pkl.load(load_path)
    md = loaded['metadata']
    if md['r'] != args.r:
        raise ValueError(f"r mismatch: expected {args.r}, found {md['r']}")
    if md['alpha'] != args.alpha:
        raise ValueError(f"alpha mismatch: expected {args.alpha}, found {md['alpha']}")
    if md['encoder'] != args.encoder:
        raise ValueError(f"Encoder mismatch: expected {args.encoder}, found {md['encoder']}")
    if list(md['params']) != list(args.params):
        raise ValueError(f"Params mismatch: expected {args.params}, found {md['params']}")
    if md['position'] != args.position:
        raise ValueError(f"Position mismatch: expected {args.position}, found {md['position']}")
    weights = loaded['weights']
    with torch.no_grad():
        for i, layer in enumerate(list_lora_layers):
            lw = weights[f'layer_{i}']
            for p in ('q', 'k', 'v', 'o'):
                name = _PROJ_NAME[p]
                if p in args.params and name in lw:
                    m = getattr(layer, name)
                    m.w_lora_A.data.copy_(torch.from_numpy(np.ascontiguousarray(lw[name]['w_lora_A'])))
                    m.w_lora_B.data.copy_(torch.from_numpy(np.ascontiguousarray(lw[name]['w_lora_B'])))
    print(f'LoRA weights loaded from {load_path}')


# ----------------------------------------------------------------------------------------------
# evaluation helpers (lora_train_vlp.py:638-666,742-811,907-919)
# ----------------------------------------------------------------------------------------------

def cls_acc(output, target, topk=1):
    """:638-644 (top-k on the GPU; ties resolved towards the smaller class index)."""
    pred = ops.topk(output.contiguous().float(), topk).long()
    correct = pred.eq(target.to(pred.device).view(-1, 1).expand_as(pred))
    return 100.0 * float(correct.float().sum().item()) / target.shape[0]


def encode_text_in_batches(clip_model, texts, batch_size=32):
    """:907-919 -- the DIFFERENTIABLE text path of the training loop (called at :975 under
    ``clip_model.train()``; the text-tower LoRA is trained through it), so no ``no_grad`` here:
    ``encode_text`` decides by ``requires_grad``.  Captions are independent rows, so the engine encodes
    them in one pass; ``batch_size`` is accepted for signature compatibility."""
    if len(texts) == 0:
        raise ValueError("No embeddings were generated. Please check your batch processing.")
    return clip_model.encode_text(clip.tokenize(list(texts)))


@torch.no_grad()
def clip_classifier(templates_dict, clip_model):
    """:647-666: per class, encode every template, L2-normalise, mean, L2-normalise -> [1, C, d]
    (one batched text pass instead of one launch chain per template)."""
    cats = list(templates_dict.keys())
    counts = {len(templates_dict[c]) for c in cats}
    if len(counts) != 1:
        raise ValueError("every class must have the same number of templates")
    t = counts.pop()
    texts = [tpl for c in cats for tpl in templates_dict[c]]
    emb = clip_model.encode_text(clip.tokenize(texts))
    w = ops.class_mean_fwd(emb.contiguous(), len(cats), t)  # [C, d]
    return w.unsqueeze(0)  # [1, C, d] like the reference's jt.stack(..., dim=1); callers .squeeze(0).t()


@torch.no_grad()
def solve_mta(image_features, text_features):
    """:742-811: image_features [V, d] unit rows (row 0 = centre view), text_features [d, C]
    -> ``mode @ text * 100`` [1, C].  One kernel, no host sync."""
    text = text_features.t().contiguous().float()
    _, logits = ops.mta(image_features.contiguous().float().unsqueeze(0), text, want_mode=False)
    return logits


@torch.no_grad()
def evaluate_views(clip_model, views, target, textual_features):
    """The per-batch body of evaluate_lora (:823-841) for ``n_img`` images at once: ``views`` [n_img, V, 3, R, R]
    (view 0 = the centre view, the rest the random crops the loader stacks behind it), ``target`` [n_img],
    ``textual_features`` [d, C] unit columns.  ONE image-tower pass over all n_img * V views, one MTA launch.
    Returns the number of correct top-1 predictions (mta, centre view, view ensemble)."""
    n_img, V = views.shape[:2]
    text_cd = textual_features.t().contiguous().float()
    feats = ops.l2norm_fwd(clip_model.encode_image(views.reshape(n_img * V, *views.shape[2:])).contiguous())
    fv = feats.reshape(n_img, V, -1)
    _, mta = ops.mta(fv, text_cd, want_mode=False)                       # solve_mta(...) = 100 * mode @ text  (:834)
    base = ops.gemm_nt(fv[:, 0].contiguous(), text_cd)                   # image_features[0] @ textual_features (:835)
    ens = ops.gemm_nt(fv.mean(dim=1).contiguous(), text_cd)              # (features @ text).mean(0) = mean(features) @ text (:837)
    tgt = target.to(mta.device).long().view(-1)
    return tuple(int((ops.topk(x.contiguous(), 1).long().view(-1) == tgt).sum().item()) for x in (mta, base, ens))


@torch.no_grad()
def evaluate_lora(args, clip_model, loader, templates=None, textual_features=None):
    """lora_train_vlp.py:813-846: MTA / centre-view / view-ensemble top-1 accuracy (in %) over a loader that yields
    ``(image [n,1,3,R,R] | [n,3,R,R], images [n,1,N,3,R,R] | [n,N,3,R,R], target, impath)`` -- the centre view and the N
    random crops of each image.  The text classifier comes from ``templates`` ({class: [captions]}, the reference reads
    them from the 'text_template' folder, :816-817) or is passed in as ``textual_features`` [d, C]."""
    clip_model.eval()
    if textual_features is None:
        if templates is None:
            raise ValueError("evaluate_lora needs `templates` ({class: [captions]}) or `textual_features` [d, C]")
        textual_features = clip_classifier(templates, clip_model).squeeze(0).t()
    dev = clip_model.device
    acc = acc1 = acc2 = 0
    tot = 0
    for batch in loader:
        image, images, target = batch[0], batch[1], batch[2]
        image = torch.as_tensor(image).to(dev).float()
        images = torch.as_tensor(images).to(dev).float()
        if image.dim() == 5:
            image = image.squeeze(1)                                     # :826
        if images.dim() == 6:
            images = images.squeeze(1)                                   # :827
        if images.dim() == 4:                                            # the reference's batch-size-1 loader: [N,3,R,R]
            images = images.unsqueeze(0)
        views = torch.cat([image.unsqueeze(1), images], dim=1)           # jt.concat((image, images)), per image (:829)
        target = torch.as_tensor(target).view(-1)
        c = evaluate_views(clip_model, views, target, textual_features)
        acc, acc1, acc2 = acc + c[0], acc1 + c[1], acc2 + c[2]
        tot += views.shape[0]
    if tot == 0:
        raise ValueError("evaluate_lora: the loader yielded no images")
    return 100.0 * acc / tot, 100.0 * acc1 / tot, 100.0 * acc2 / tot


# ----------------------------------------------------------------------------------------------
# flat trainable buffer + stage-1 step (lora_train_vlp.py:946,956-1002)
# ----------------------------------------------------------------------------------------------

class FlatTrainables:
    """Re-homes every stacked LoRA tensor (text blocks, then vision blocks: apply_lora order) plus the
    optional prompt / VPT tokens into ONE contiguous fp32 buffer with matching gradient and AdamW
    moment buffers: a single RCCL all-reduce and a single optimiser launch per step."""

    def __init__(self, model, extra: Sequence[nn.Parameter] = ()):
        self.model = model
        entries = []
        for tower in (model.transformer, model.visual.transformer):
            for blk in tower.resblocks:
                a = blk.attn
                if getattr(a, "is_lora_mha", False):
                    for name, p, _ in a.stacked():
                        entries.append((a, name, p))
        n = sum(p.numel() for _, _, p in entries) + sum(p.numel() for p in extra)
        if n == 0:
            raise ValueError("model has no trainable adapter parameters (call apply_lora first)")
        dev = model.device
        self.params = torch.zeros(n, device=dev)
        self.grads = torch.zeros(n, device=dev)
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        off = 0
        for a, name, p in entries:
            k = p.numel()
            self.params[off:off + k].copy_(p.reshape(-1))
            a.rebind(name, self.params[off:off + k].view_as(p), self.grads[off:off + k].view_as(p))
            off += k
        self.extra = []
        for p in extra:
            k = p.numel()
            self.params[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.params[off:off + k].view_as(p)
            p.grad_slot = self.grads[off:off + k].view_as(p)
            self.extra.append(p)
            off += k
        self.numel = n
        model.invalidate_engine()

    def zero_grad(self):
        self.grads.zero_()


class LoRATrainer:
    """One object = the state of run_lora (:921-1023) that matters for the hot path: model with adapters,
    flat trainables, AdamW moments, step counter.  ``step`` is the loop body :956-1002:

        text features of every class (with grad)  ->  per class normalise/mean/normalise
        image features of the batch               ->  normalise
        logits = 100 * img @ txt^T ; mean cross entropy ; backward into LoRA (+ prompt) ; AdamW

    Data-parallel: every rank holds the full model, takes its shard of the image batch, and the flat
    gradient buffer is summed with one all-reduce (losses are pre-scaled by B_local / B_global)."""

    def __init__(self, model, lr: float = 2e-4, weight_decay: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-8,
                 logit_scale: float = 100.0, prompt_ctx: Optional[nn.Parameter] = None, process_group=None,
                 shard_text: bool = True):
        from clipfs import dist as D
        self.model = model
        extra = []
        if prompt_ctx is not None:
            extra.append(prompt_ctx)
        if model.visual.VPT is not None and model.visual.VPT.requires_grad:
            extra.append(model.visual.VPT)
        self.flat = FlatTrainables(model, extra)
        self.prompt_ctx = prompt_ctx
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.logit_scale = logit_scale
        self.t = 0
        self.pg = process_group
        self.rank, self.world = D.world_info(process_group)
        self.shard_text = shard_text and (self.world > 1 or D.FORCE_COLLECTIVES)
        self.overlap_towers = True  # text tower on a side HIP stream (set False to serialise, e.g. for per-kernel timing)
        # class-sharded text tower: fixed exchange buffers (allocated on first use, never re-zeroed: padding rows of
        # the send block / the gradient table are written by nobody and stay zero)
        self._xbuf = {}
        self.collectives_per_step = (3 if self.shard_text else 1) if (self.world > 1 or D.FORCE_COLLECTIVES) else 0
        self.time_collectives = False      # bench: bracket every collective with HIP events on the launch stream
        self._coll_events = []

    # -- collectives ---------------------------------------------------------------------------------
    def _exchange_buffers(self, classes: int, width: int):
        from clipfs import dist as D
        key = (classes, width)
        b = self._xbuf.get(key)
        if b is None:
            s = D.block_rows(classes, self.world)
            dev = self.flat.params.device
            b = dict(S=s, send=torch.zeros(s, width, device=dev), full=torch.empty(self.world * s, width, device=dev),
                     dfull=torch.zeros(self.world * s, width, device=dev), dmine=torch.empty(s, width, device=dev))
            self._xbuf = {key: b}
        return b

    def _timed(self, name, fn):
        if not self.time_collectives:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self._coll_events.append((name, e0, e1))
        return out

    def collective_times_ms(self):
        """{name: [ms, ...]} of the collectives bracketed since the last call (needs a prior synchronize)."""
        out = {}
        for name, e0, e1 in self._coll_events:
            out.setdefault(name, []).append(e0.elapsed_time(e1))
        self._coll_events = []
        return out

    def forward_backward(self, images, captions, target, templates_per_class: int = 1, global_batch: Optional[int] = None,
                         row_offset: int = 0):
        """``images`` / ``target`` are THIS RANK's shard of the batch (rows ``row_offset ...`` of the global batch),
        ``captions`` the full caption table [C * t, 77] (class major).  Accumulates gradients into the flat buffer;
        returns (loss_sum_local [1], correct_local [1], logits_local [B_local, C]).

        Collectives (world > 1, class-sharded text): all_gather of the class-feature block, reduce_scatter of its
        gradient (here), all_reduce of the flat gradient (optimizer_step) -- clipfs/dist.py."""
        from clipfs import dist as D
        m = self.model
        eng = m.engine
        seed = eng.next_seed() if m.training else 0
        B = images.shape[0]
        gb = global_batch or B * self.world
        t = templates_per_class
        classes = captions.shape[0] // t
        d = m.embed_dim
        c_lo, c_hi = D.block_bounds(classes, self.rank, self.world) if self.shard_text else (0, classes)
        if self.shard_text and eng.trim_text and m.training and eng.txt.lora_dropout_rate() > 0:
            # dropout masks are indexed by token row = caption * seq + position; trimming makes `seq` the local shard's
            # last EOT, so ranks would index different (and overlapping) rows than the one-process run
            raise ValueError("trim_text cannot be combined with a class-sharded text tower and LoRA dropout > 0: the "
                             "Philox rows would depend on each rank's trimmed length (use trim_text=False or shard_text=False)")
        xb = self._exchange_buffers(classes, d) if self.shard_text else None
        # The two towers are independent until the logits: the text tower runs on a side HIP stream so that
        # its kernels fill the CUs the image tower's launches leave idle (small per-rank batches) and vice versa.
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap_towers else main
        side.wait_stream(main)
        emb = tctx = txt = None
        with torch.cuda.stream(side):
            if c_hi > c_lo:
                # dropout masks are indexed by GLOBAL rows (row0): a sharded run draws the masks of the one-process run
                emb, tctx = eng.text_forward(captions[c_lo * t:c_hi * t], self.prompt_ctx, True, seed, row0=c_lo * t)
                txt = ops.class_mean_fwd(emb, c_hi - c_lo, t, out=None if xb is None else xb["send"][:c_hi - c_lo])
        feat, ictx = eng.vit_forward(images, True, seed, row0=row_offset)
        img_n, inv = ops.l2norm_fwd(feat, save_inv=True)
        main.wait_stream(side)
        if self.shard_text:
            full = self._timed("all_gather", lambda: D.allgather_blocks(xb["send"], xb["full"], self.pg))
            txt = full[:classes]
        logits = ops.gemm_nt(img_n, txt, alpha=self.logit_scale)
        self.last_features = (img_n, txt)  # unit image features of this rank's shard, unit class features [C, d] (tests)
        loss_sum, dl, correct = ops.cross_entropy(logits, target, True, grad_scale=B / gb)
        d_img_n = ops.matmul_small(dl, txt, B, d, classes, classes, 1, d, 1, self.logit_scale)
        d_txt = ops.matmul_small(dl, img_n, classes, d, B, 1, classes, d, 1, self.logit_scale,
                                 out=None if xb is None else xb["dfull"][:classes])
        if self.shard_text:  # every rank needs the batch-total gradient of ITS classes only
            mine = self._timed("reduce_scatter", lambda: D.reduce_scatter_blocks(xb["dfull"], xb["dmine"], self.pg))
            d_txt_local = mine[:c_hi - c_lo]
        else:
            d_txt_local = d_txt
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if c_hi > c_lo:
                d_emb = ops.class_mean_bwd(emb, d_txt_local, c_hi - c_lo, t)
                slot = None if self.prompt_ctx is None else self.prompt_ctx.grad_slot
                eng.text_backward(tctx, d_emb, slot)
        eng.vit_backward(ictx, ops.l2norm_bwd(d_img_n, img_n, inv))
        main.wait_stream(side)
        return loss_sum, correct, logits

    def _side_stream(self):
        # one per device for everything in the process (clipfs/streams.py: HIP maps streams onto a handful of hardware
        # queues, and a fresh stream per user ends up sharing a queue with another one)
        from clipfs import streams
        return streams.side_stream(self.model.device)

    def optimizer_step(self):
        from clipfs import dist as D
        grad_scale = 1.0
        if self.world > 1 or D.FORCE_COLLECTIVES:
            # replicated text tower (shard_text=False): every rank back-propagated the text gradient of ITS images
            # only, so this same sum is already the batch total (d_txt is linear in the local dlogits)
            self._timed("all_reduce", lambda: D.allreduce_sum_(self.flat.grads, self.pg))
        self.t += 1
        ops.adamw(self.flat.params, self.flat.grads, self.flat.m, self.flat.v, self.t, self.lr, self.betas, self.eps,
                  self.wd, grad_scale)

    def step(self, images, captions, target, templates_per_class: int = 1, global_batch: Optional[int] = None,
             row_offset: int = 0):
        self.flat.zero_grad()
        out = self.forward_backward(images, captions, target, templates_per_class, global_batch, row_offset)
        self.optimizer_step()
        return out
