"""Stage-2 pieces of the reference's ``slow_pace.py`` that sit on the hot path, on the HIP engine:

    VLPromptLearner   slow_pace.py:110-205   4 shared learnable text-prompt tokens (ctx)
    TextEncoder       slow_pace.py:828-848   text tower fed by prompt embeddings
    Channel_LP        slow_pace.py:1195-1206 the repo's "LP++" head (per-channel affine + Linear(512,403))
    logit_normalize   slow_pace.py:1276-1280
    solve_mta         slow_pace.py:1363-1433 MTA returning the MODE feature [1, d]

The stage-2 loss zoo (SCL / KL / MoCo, slow_pace.py:1479-1716) is out of scope.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import nn

from clipfs import engine as E
from clipfs import ops
from jclip import clip
from lora_train_vlp import clip_classifier  # noqa: F401  (re-exported like the reference's copy)


class PromptBatch:
    """What VLPromptLearner.execute hands to TextEncoder: the reference concatenates
    [SOT emb | ctx | suffix emb] into a [C, 77, d] tensor (slow_pace.py:185-199); the engine keeps the
    three pieces symbolic (token ids + the shared ctx rows) and assembles them inside the embedding
    kernel, so nothing of size C*77*d is materialised or differentiated through."""

    def __init__(self, ctx: nn.Parameter, tokenized_prompts: torch.Tensor):
        self.ctx = ctx
        self.tokenized_prompts = tokenized_prompts

    def materialize(self, model) -> torch.Tensor:
        emb = model.token_embedding.weight.data[self.tokenized_prompts.to(model.device)]
        n = self.ctx.shape[0]
        return torch.cat([emb[:, :1], self.ctx.data.unsqueeze(0).expand(emb.shape[0], -1, -1), emb[:, 1 + n:]], dim=1)


class VLPromptLearner(nn.Module):
    """slow_pace.py:110-205.  ``clip_zs`` (the frozen zero-shot CLIP used for ``fixed_embeddings``) is optional."""

    def __init__(self, classnames: Sequence[str], clip_model, clip_zs=None, templates: Optional[dict] = None):
        super().__init__()
        n_ctx = 4
        ctx_init = "a photo of a"
        if clip_model.visual.input_resolution != 224:
            raise AssertionError(f"cfg_imsize (224) must equal to clip_imsize ({clip_model.visual.input_resolution})")
        prompt = clip.tokenize(ctx_init)
        emb = clip_model.token_embedding(prompt)          # [1, 77, d]
        self.ctx = nn.Parameter(emb[0, 1:1 + n_ctx, :].clone().contiguous())   # :124-131
        classnames = [name.replace("_", " ") for name in classnames]                # :145
        self.name_lens = [len(clip._tok().encode(name)) for name in classnames]
        prompts = [ctx_init + " " + name + "." for name in classnames]              # :147
        self.tokenized_prompts = clip.tokenize(prompts).to(clip_model.device)       # (n_cls, n_tkn)
        self.n_cls = len(classnames)
        self.n_ctx = n_ctx
        self.fixed_embeddings = None
        if clip_zs is not None and templates is not None:
            self.fixed_embeddings = clip_classifier(templates, clip_zs).squeeze(0)  # :163
        self._model = [clip_model]  # not registered as a sub-module

    @property
    def token_prefix(self):
        return self._model[0].token_embedding.weight.data[self.tokenized_prompts[:, :1]]

    @property
    def token_suffix(self):
        return self._model[0].token_embedding.weight.data[self.tokenized_prompts[:, 1 + self.n_ctx:]]

    def forward(self) -> PromptBatch:
        return PromptBatch(self.ctx, self.tokenized_prompts)

    execute = forward


class TextEncoder(nn.Module):
    """slow_pace.py:828-848: ``TextEncoder(clip_model)(prompts, tokenized_prompts)`` -> [C, E];
    differentiable with respect to the prompt ctx and the text-tower LoRA parameters."""

    def __init__(self, clip_model):
        super().__init__()
        self._model = [clip_model]
        self.dtype = clip_model.dtype

    def forward(self, prompts, tokenized_prompts=None):
        model = self._model[0]
        if isinstance(prompts, PromptBatch):
            ids = prompts.tokenized_prompts if tokenized_prompts is None else tokenized_prompts
            return E.encode_text(model, ids, prompts.ctx)
        raise TypeError("TextEncoder expects the PromptBatch returned by VLPromptLearner() "
                        "(raw [C,77,d] embedding tensors are not differentiated through on the HIP path)")

    execute = forward


class _ChannelLPFn(torch.autograd.Function):
    """z = (scale1 * f + bias1) W^T + c with gradients for scale1, bias1, W, c (and f)."""

    @staticmethod
    def forward(ctx, f, scale1, bias1, w, c):
        f = f.contiguous()
        fp = ops.channel_affine(f, scale1.contiguous(), bias1.contiguous())
        ctx.save_for_backward(f, fp, scale1, w)
        return ops.gemm_nt(fp, w.contiguous(), bias=c.contiguous())

    @staticmethod
    def backward(ctx, dz):
        f, fp, scale1, w = ctx.saved_tensors
        dz = dz.contiguous()
        R, Cn = dz.shape
        d = f.shape[1]
        dw = ops.matmul_small(dz, fp, Cn, d, R, 1, Cn, d, 1)          # dW[c,k] = sum_r dz[r,c] f'[r,k]
        dc = ops.colsum(dz)
        dfp = ops.matmul_small(dz, w.contiguous(), R, d, Cn, Cn, 1, d, 1)  # df'[r,k] = sum_c dz[r,c] W[c,k]
        ds = ops.colsum(dfp, f)
        db = ops.colsum(dfp)
        df = ops.channel_affine(dfp, scale1.contiguous(), torch.zeros_like(scale1)) if ctx.needs_input_grad[0] else None
        return df, ds, db, dw, dc


class _LogitNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        z = z.contiguous()
        ctx.save_for_backward(z)
        return ops.logit_normalize(z)

    @staticmethod
    def backward(ctx, dzn):
        (z,) = ctx.saved_tensors
        return ops.logit_normalize_bwd(z, dzn.contiguous())


class Channel_LP(nn.Module):
    """slow_pace.py:1195-1206: out = Linear_{512->403}(scale1 * f + bias1); ``fc.weight`` is initialised from the
    zero-shot text features by the caller (:1537-1540).  Differentiable with respect to scale1 / bias1 / fc (the
    stage-2 head training, :1671-1675); forward and backward are HIP kernels."""

    def __init__(self, in_dim: int = 512, n_classes: int = 403, device=None):
        super().__init__()
        device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
        self.scale1 = nn.Parameter(torch.ones(in_dim, device=device))
        self.bias1 = nn.Parameter(torch.zeros(in_dim, device=device))
        fc = nn.Linear(in_dim, n_classes)
        self.fc = fc.to(device) if device is not None else fc

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        f = features.to(self.scale1.device, torch.float32)
        return _ChannelLPFn.apply(f, self.scale1, self.bias1, self.fc.weight, self.fc.bias)

    execute = forward


def logit_normalize(logit: torch.Tensor) -> torch.Tensor:
    """slow_pace.py:1276-1280 (differentiable)."""
    return _LogitNormFn.apply(logit.float())


@torch.no_grad()
def solve_mta(image_features: torch.Tensor, text_features: torch.Tensor) -> torch.Tensor:
    """slow_pace.py:1363-1433: returns the mode feature [1, d] (the stage-1 / ood variant returns logits)."""
    text = text_features.t().contiguous().float()
    mode, _ = ops.mta(image_features.contiguous().float().unsqueeze(0), text, want_logits=False)
    return mode
