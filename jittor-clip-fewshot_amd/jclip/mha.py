"""``MultiheadAttention`` parameter holder with the attribute surface the reference's adapter code
touches (jclip/mha.py:469-556: ``embed_dim, kdim, vdim, _qkv_same_embed_dim, num_heads, batch_first,
head_dim, in_proj_weight, in_proj_bias, out_proj``).  ``apply_lora`` keys on the CLASS NAME
``MultiheadAttention`` (lora_train_vlp.py:526), so the name is part of the contract.

Inside a transformer tower the arithmetic is not run module by module: the C++ tower driver
(csrc/tower.hip) sequences the fused HIP kernels for all blocks.  ``forward`` below exists for callers
that invoke the module directly (sequence-first input like the reference, inference only)."""
from __future__ import annotations

import torch
from torch import nn


class _OutProj(nn.Module):
    def __init__(self, weight: torch.Tensor, bias: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=False)
        self.bias = nn.Parameter(bias, requires_grad=False)
        self.in_features = weight.shape[1]
        self.out_features = weight.shape[0]


class MultiheadAttention(nn.Module):
    __constants__ = ["batch_first"]

    def __init__(self, embed_dim: int, num_heads: int, in_proj_weight: torch.Tensor = None,
                 in_proj_bias: torch.Tensor = None, out_proj_weight: torch.Tensor = None,
                 out_proj_bias: torch.Tensor = None, device=None):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim = embed_dim
        self.kdim = embed_dim
        self.vdim = embed_dim
        self._qkv_same_embed_dim = True
        self.num_heads = num_heads
        self.dropout = 0.0
        self.batch_first = False
        self.head_dim = embed_dim // num_heads
        if self.head_dim != 64:
            raise ValueError("the HIP attention kernels are specialised for head_dim 64 (all CLIP ViT/text towers)")
        z = lambda *s: torch.zeros(*s, device=device, dtype=torch.float32)
        self.in_proj_weight = nn.Parameter(in_proj_weight if in_proj_weight is not None else z(3 * embed_dim, embed_dim),
                                           requires_grad=False)
        self.in_proj_bias = nn.Parameter(in_proj_bias if in_proj_bias is not None else z(3 * embed_dim),
                                         requires_grad=False)
        self.out_proj = _OutProj(out_proj_weight if out_proj_weight is not None else z(embed_dim, embed_dim),
                                 out_proj_bias if out_proj_bias is not None else z(embed_dim))

    @torch.no_grad()
    def forward(self, query, key=None, value=None, need_weights=False, attn_mask=None, **_):
        """[L, N, d] in, ([L, N, d], None) out; causal iff an attn_mask is given (CLIP only uses the
        causal mask, jclip/model.py:189-193)."""
        from clipfs import ops
        if need_weights:
            raise NotImplementedError("attention weights are never materialised (need_weights=False only)")
        L, N, d = query.shape
        x = query.permute(1, 0, 2).reshape(N * L, d).contiguous()
        qkv = ops.gemm_nt(x, self.in_proj_weight.data, bias=self.in_proj_bias.data)
        o = ops.attention_fwd(qkv, N, L, self.num_heads, attn_mask is not None)
        y = ops.gemm_nt(o, self.out_proj.weight.data, bias=self.out_proj.bias.data)
        return y.reshape(N, L, d).permute(1, 0, 2).contiguous(), None

    execute = forward
