"""Stage-2 pieces of the reference's ``slow_pace.py`` that sit on the hot path, on the HIP engine:

    VLPromptLearner   slow_pace.py:110-205   4 shared learnable text-prompt tokens (ctx)
    TextEncoder       slow_pace.py:828-848   text tower fed by prompt embeddings
    Channel_LP        slow_pace.py:1195-1206 the repo's "LP++" head (per-channel affine + Linear(512,403))
    logit_normalize   slow_pace.py:1276-1280
    solve_mta         slow_pace.py:1363-1433 MTA returning the MODE feature [1, d]

    l1_loss / kl_div  slow_pace.py:1653-1658,1170-1177  the SCL self-consistency terms
    CosineAnnealingLR slow_pace.py:1591      jt.lr_scheduler.CosineAnnealingLR(total_epoch, eta_min=1e-6)
    stage2_loss       slow_pace.py:1622-1688 the stage-2 objective without the MoCo branch
    Stage2Trainer     slow_pace.py:1590-1697 the loop body (prompt ctx + VPT + head, AdamW + cosine LR)
    load_lora_swa     slow_pace.py:736-816   average of several saved LoRA files
    pre_load_zs       slow_pace.py:1435-1477 cached zero-shot MTA features of the training images
    PromptQueue       README.md:22           queue of learned prompt features blended with the hand-written ones

    load_moco / Moco_Adapter / pre_load_features_moco / moco_adapter_init / tfm_moco / tfm_clip
                      slow_pace.py:1208-1219,1237-1274,1151-1168,1542-1552 the MoCo-v3 ResNet-50 auxiliary branch
                      (frozen extractor on the HIP GEMM, clipfs/resnet.py; ``stage2_loss(..., moco_features=...)`` adds
                      ``loss_aux`` :1677-1680)
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

    # -- Jittor Module.save / Module.load (slow_pace.py:1712, test.py:1821) ------------------------------------------
    def save(self, path: str) -> None:
        """``prompt_learner.save('test_pkl/PromptLearner.pkl')``: the learnable ``ctx`` plus the tensors the reference's
        module also holds as Vars (token_prefix / token_suffix / tokenized_prompts / fixed_embeddings), so that the
        reference can read the file back."""
        from clipfs import module_io
        extra = [("token_prefix", self.token_prefix), ("token_suffix", self.token_suffix),
                 ("tokenized_prompts", self.tokenized_prompts)]
        if self.fixed_embeddings is not None:
            extra.append(("fixed_embeddings", self.fixed_embeddings))
        module_io.save_module(self, path, extra)

    def load(self, path: str) -> None:
        """Only ``ctx`` is state here: prefix / suffix embeddings are re-derived from the model's token embedding."""
        from clipfs import module_io
        module_io.load_module(self, path, ignore=("token_prefix", "token_suffix", "tokenized_prompts", "fixed_embeddings"))


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

    def save(self, path: str) -> None:
        """``channel_lp.save('test_pkl/channel.pkl')`` (slow_pace.py:1709): {scale1, bias1, fc.weight, fc.bias}."""
        from clipfs import module_io
        module_io.save_module(self, path)

    def load(self, path: str) -> None:
        """``channel_lp.load('test_pkl/channel.pkl')`` (test.py:1819)."""
        from clipfs import module_io
        module_io.load_module(self, path)


def logit_normalize(logit: torch.Tensor) -> torch.Tensor:
    """slow_pace.py:1276-1280 (differentiable)."""
    return _LogitNormFn.apply(logit.float())


def load_lora_swa(args, list_lora_layers, swa_weights_folder: str) -> int:
    """slow_pace.py:736-816: element-wise average of every ``save_lora`` file in ``swa_weights_folder`` (sub-folders
    skipped, any metadata mismatch -> ValueError, missing folder -> FileNotFoundError; the reference's metadata check
    reads ``args.alpha_lora``, falling back here to ``args.alpha`` which the rest of the API uses), written into the
    adapters.  Files are read with the inert pickle reader; the average is accumulated in float64 in directory-listing
    order and rounded once.  Returns the number of files averaged."""
    import os

    import numpy as np

    from clipfs import safe_pkl
    from lora_train_vlp import _PROJ_NAME
    if not os.path.exists(swa_weights_folder):
        raise FileNotFoundError(f"folder {swa_weights_folder} does not exist.")
    acc, count = None, 0
    alpha = getattr(args, "alpha_lora", getattr(args, "alpha", None))
    for filename in os.listdir(swa_weights_folder):
        path = os.path.join(swa_weights_folder, filename)
        if os.path.isdir(path):
            continue
        loaded = safe_pkl.load(path)
        md = loaded["metadata"]
        for key, want in (("r", args.r), ("alpha", alpha), ("encoder", args.encoder), ("position", args.position)):
            if md[key] != want:
                raise ValueError(f"{key} mismatch: expected {want}, found {md[key]}")
        if list(md["params"]) != list(args.params):
            raise ValueError(f"params mismatch: expected {args.params}, found {md['params']}")
        weights = loaded["weights"]
        if acc is None:
            acc = {ln: {pn: {k: np.zeros(np.shape(v), np.float64) for k, v in ab.items()} for pn, ab in lw.items()}
                   for ln, lw in weights.items()}
        for ln, lw in weights.items():
            for pn, ab in lw.items():
                for k, v in ab.items():
                    acc[ln][pn][k] += np.asarray(v, np.float64)
        count += 1
    if count == 0:
        raise ValueError(f"no LoRA files in {swa_weights_folder}")
    with torch.no_grad():
        for i, layer in enumerate(list_lora_layers):
            lw = acc[f"layer_{i}"]
            for p in ("q", "k", "v", "o"):
                name = _PROJ_NAME[p]
                if p in args.params and name in lw:
                    m = getattr(layer, name)
                    m.w_lora_A.data.copy_(torch.from_numpy((lw[name]["w_lora_A"] / count).astype(np.float32)))
                    m.w_lora_B.data.copy_(torch.from_numpy((lw[name]["w_lora_B"] / count).astype(np.float32)))
    return count


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        loss, da = ops.l1_loss(a, b, want_grad=True)
        ctx.save_for_backward(da)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (da,) = ctx.saved_tensors
        return da * g, None


def l1_loss(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """jittor ``nn.l1_loss``: mean |output - target| (slow_pace.py:1654-1655); gradient flows to ``output`` only (the
    zero-shot side is a constant there)."""
    return _L1Fn.apply(output.float(), target.detach().float())


class _KLLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target_logits, denom):
        rows, dx = ops.kl_logits(logits, target_logits, want_grad=True, grad_scale=1.0 / denom)
        ctx.save_for_backward(dx)
        return rows.sum() / denom

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None, None


def scl_logits_loss(cosine_similarity: torch.Tensor, zero_shot_logits: torch.Tensor) -> torch.Tensor:
    """slow_pace.py:1656-1658: ``kl_div(log_softmax(cos), log_softmax(zs), 'sum') / cos.numel()`` with ``kl_div`` of
    :1170-1177 (= sum exp(b) (b - a)); the zero-shot logits carry no gradient (:1651 ``jt.no_grad``)."""
    return _KLLogitsFn.apply(cosine_similarity.float(), zero_shot_logits.detach().float(), float(cosine_similarity.numel()))


class CosineAnnealingLR:
    """``jt.lr_scheduler.CosineAnnealingLR(optimizer, T_max, eta_min)`` as slow_pace.py:1591,1697 uses it: after
    ``step()`` number ``t`` the learning rate is ``eta_min + (base_lr - eta_min) * (1 + cos(pi * t / T_max)) / 2``."""

    def __init__(self, base_lr: float, T_max: int, eta_min: float = 1e-6):
        self.base_lr, self.T_max, self.eta_min = float(base_lr), int(T_max), float(eta_min)
        self.last_epoch = 0

    def get_lr(self) -> float:
        import math
        return self.eta_min + (self.base_lr - self.eta_min) * (1.0 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2.0

    def step(self) -> float:
        self.last_epoch += 1
        return self.get_lr()


class _LinearFn(torch.autograd.Function):
    """z = f W^T + c with gradients for W and c (and f): the Moco_Adapter head."""

    @staticmethod
    def forward(ctx, f, w, c):
        f = f.contiguous()
        ctx.save_for_backward(f, w)
        return ops.gemm_nt(f, w.contiguous(), bias=c.contiguous())

    @staticmethod
    def backward(ctx, dz):
        f, w = ctx.saved_tensors
        dz = dz.contiguous()
        R, Cn = dz.shape
        d = f.shape[1]
        dw = ops.matmul_small(dz, f, Cn, d, R, 1, Cn, d, 1)
        dc = ops.colsum(dz)
        df = ops.matmul_small(dz, w.contiguous(), R, d, Cn, Cn, 1, d, 1) if ctx.needs_input_grad[0] else None
        return df, dw, dc


class Moco_Adapter(nn.Module):
    """slow_pace.py:1208-1219: ``fc = Linear(2048, 403)`` on the frozen ResNet-50 features; trained by ``loss_aux``."""

    def __init__(self, in_dim: int = 2048, n_classes: int = 403, device=None):
        super().__init__()
        device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
        fc = nn.Linear(in_dim, n_classes)
        self.fc = fc.to(device) if device is not None else fc

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        return _LinearFn.apply(features.to(self.fc.weight.device, torch.float32), self.fc.weight, self.fc.bias)

    execute = forward

    def save(self, path: str) -> None:
        """``moco_adapter.save('test_pkl/moco_adapter.pkl')`` (slow_pace.py:1710)."""
        from clipfs import module_io
        module_io.save_module(self, path)

    def load(self, path: str) -> None:
        from clipfs import module_io
        module_io.load_module(self, path)


CLIP_MEAN_STD = ((0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711))
MOCO_MEAN_STD = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))


def _normalizer(mean, std):
    def tfm(images: torch.Tensor) -> torch.Tensor:
        """[B, 3, H, W] in [0, 1] -> (x - mean) / std per channel (jittor.transform.ImageNormalize)."""
        m = torch.tensor(mean, device=images.device, dtype=torch.float32).view(1, 3, 1, 1)
        s = torch.tensor(std, device=images.device, dtype=torch.float32).view(1, 3, 1, 1)
        return (images.float() - m) / s
    return tfm


tfm_clip = _normalizer(*CLIP_MEAN_STD)   # slow_pace.py:1273
tfm_moco = _normalizer(*MOCO_MEAN_STD)   # slow_pace.py:1274


def load_moco(pretrain_path, device=None):
    """slow_pace.py:1237-1271: ``(model, 2048)``; the checkpoint is ``{'state_dict': {'base_encoder.X': array, ...}}``
    (a pickle of numpy arrays written by the reference's pth_to_pkl.py; read with the inert reader) or such a dict
    itself.  ``FileNotFoundError`` when the file is missing, like the reference."""
    import os

    import numpy as np

    from clipfs import resnet, safe_pkl
    if isinstance(pretrain_path, dict):
        ck = pretrain_path
    else:
        if not os.path.isfile(pretrain_path):
            print("=> no checkpoint found at '{}'".format(pretrain_path))
            raise FileNotFoundError(pretrain_path)
        ck = safe_pkl.load(pretrain_path)
    sd = ck["state_dict"] if "state_dict" in ck else ck
    sd = {k: torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v for k, v in sd.items()}
    sd = resnet.strip_moco_prefix(sd)
    device = device or torch.device("cuda", torch.cuda.current_device())
    return resnet.MocoResNet50(sd, device), 2048


@torch.no_grad()
def pre_load_features_moco(moco_model, loader):
    """slow_pace.py:1151-1168: unit-norm ResNet features of every training image and their labels; ``loader`` yields
    ``(images in [0, 1], target, index)``."""
    feats, labels = [], []
    for batch in loader:
        images, target = batch[0], batch[1]
        feats.append(moco_model(tfm_moco(torch.as_tensor(images).to(moco_model.device))))
        labels.append(torch.as_tensor(target).view(-1))
    f = torch.cat(feats, dim=0)
    f = ops.l2norm_fwd(f.contiguous())            # mean over ONE augmentation pass (:1155) is the identity
    return f, torch.cat(labels).to(f.device)


@torch.no_grad()
def moco_adapter_init(moco_adapter: "Moco_Adapter", moco_features: torch.Tensor, moco_labels: torch.Tensor) -> None:
    """slow_pace.py:1545-1551: ``fc.weight[label] = sum of the unit features of that class`` (few-shot prototype init)."""
    w = torch.zeros_like(moco_adapter.fc.weight)
    w.index_add_(0, moco_labels.to(w.device).long(), moco_features.to(w.device, torch.float32))
    moco_adapter.fc.weight.copy_(w)


def stage2_loss(image_features: torch.Tensor, text_features: torch.Tensor, target: torch.Tensor,
                zs_image_features: torch.Tensor, zs_text_features: torch.Tensor, channel_lp: "Channel_LP",
                lp_image_features: torch.Tensor, lp_text_features: torch.Tensor,
                moco_adapter: Optional["Moco_Adapter"] = None, moco_features: Optional[torch.Tensor] = None):
    """The stage-2 objective of slow_pace.py:1636-1688.  With ``moco_adapter`` and ``moco_features`` (= ``moco_model(
    tfm_moco(images))``, [B, 2048], no grad) the MoCo term ``loss_aux = CE(logit_normalize(moco_adapter(features)),
    target)`` (:1677-1680) is added as in :1688; without them the objective is ``sim_ce + L_SCL + lp_ce``.

    ``image_features`` [B, d] / ``text_features`` [C, d]: the prompted, UN-normalised tower outputs (with grad);
    ``zs_*``: the cached zero-shot features (unit norm, constants; :1650,1654-1655);
    ``lp_image_features`` [B, d]: raw image features of a second, no-grad forward (:1660-1661), ``lp_text_features``
    [C, d]: one of the cached zero-shot template sets (:1662) -- the Channel_LP head is trained on their concatenation
    with targets ``concat(target, arange(C))`` (:1663-1669).
    Returns (loss, dict of the terms, cosine_similarity)."""
    img = E.l2_normalize(image_features)
    txt = E.l2_normalize(text_features)
    cos = E.cosine_logits(img, txt, 100.0)                                   # :1640
    with torch.no_grad():
        zs_logits = E.cosine_logits(zs_image_features.float(), zs_text_features.float(), 100.0)  # :1650
    loss_scl_text = l1_loss(txt, zs_text_features)                           # :1654
    loss_scl_image = l1_loss(img, zs_image_features)                         # :1655
    loss_scl_logits = scl_logits_loss(cos, zs_logits)                        # :1656-1658
    feats = torch.cat((lp_image_features.detach().float(), lp_text_features.detach().float()), dim=0)  # :1663
    out_lp = logit_normalize(channel_lp(feats))                              # :1665-1666
    C = text_features.shape[0]
    tgt_lp = torch.cat((target.to(out_lp.device).long(), torch.arange(C, device=out_lp.device)))  # :1667-1668
    lp_ce = E.cross_entropy_loss(out_lp, tgt_lp)                             # :1669
    sim_ce = E.cross_entropy_loss(cos, target)                               # :1686
    l_scl = loss_scl_logits + loss_scl_text + loss_scl_image                 # :1684
    loss = sim_ce + l_scl + lp_ce
    terms = {"sim_ce": sim_ce, "scl_text": loss_scl_text, "scl_image": loss_scl_image, "scl_logits": loss_scl_logits,
             "lp_ce": lp_ce}
    if moco_adapter is not None and moco_features is not None:
        out_moco = logit_normalize(moco_adapter(moco_features.detach()))     # :1678-1679
        loss_aux = E.cross_entropy_loss(out_moco, target)                    # :1680
        loss = loss + loss_aux                                               # :1688
        terms["loss_aux"] = loss_aux
    return loss, terms, cos


@torch.no_grad()
def pre_load_zs(clip_model_zs, views: torch.Tensor, text_features_zs: torch.Tensor) -> torch.Tensor:
    """slow_pace.py:1435-1454 (and its twin :1456-1477): the cached zero-shot MTA feature of every training image --
    ``views`` [n_img, V, 3, R, R] (view 0 = the centre preprocess, the rest the loader's random crops; build them on the
    GPU with ``tta.make_tta_views``), ``text_features_zs`` [C, d] the zero-shot classifier.  Per image: encode the V views,
    L2-normalise, ``solve_mta`` -> the mode feature (:1445-1449).  Returns [n_img, d] (the reference pickles the list as
    ``features_zs1.pkl``; Stage2Trainer takes the tensor directly)."""
    import ood
    _, mode = ood.mta_scores(clip_model_zs, views, text_features_zs, want_mode=True)
    return mode


class PromptQueue:
    """README.md:22 ("learning-pace control"): every N epochs the learned prompt-derived class features are pushed into
    a fixed-length queue (the oldest entry drops out) and the queue is averaged with the hand-written / LLM-generated
    prompt features.  The reference ships the description but not this code (its eval path fuses logits of the prompt and
    hand-written classifiers instead, test.py:1729-1742): this is the build's restatement, kept deliberately minimal --
    ``push`` stores unit-norm [C, d] text features, ``blend`` returns normalise(w * mean(queue) + (1 - w) * hand)."""

    def __init__(self, maxlen: int = 4):
        from collections import deque
        self.q = deque(maxlen=maxlen)

    def push(self, text_features: torch.Tensor) -> None:
        self.q.append(E.l2_normalize(text_features.detach().float()).clone())

    def __len__(self) -> int:
        return len(self.q)

    @torch.no_grad()
    def blend(self, hand_features: torch.Tensor, weight: float = 0.5) -> torch.Tensor:
        hand = E.l2_normalize(hand_features.float())
        if not self.q:
            return hand
        mean = torch.stack(list(self.q)).mean(0)
        return E.l2_normalize(weight * mean + (1.0 - weight) * hand)


class Stage2Trainer:
    """The loop body of slow_pace.py:1622-1697 as one object: prompt ctx + VPT tokens + Channel_LP head (+ the
    Moco_Adapter when ``moco_model`` / ``moco_adapter`` are given: ``loss_aux`` :1677-1680, parameters :1584-1586) are trained (the LoRA adapters stay applied but frozen: :1551-1556 clears ``requires_grad`` on
    everything of the CLIP model that is not a VPT parameter), AdamW(weight_decay 1e-2, betas (0.9, 0.999)) with
    ``CosineAnnealingLR(total_epoch, eta_min=1e-6)`` stepped once per iteration (:1590-1591,1696-1697).

    ``zs_image_features`` [N, d]: cached unit-norm zero-shot image features of the training set (``features_zs1.pkl``
    in the reference), indexed by the loader's sample index; ``zs_text_features`` [C, d]: zero-shot classifier;
    ``zs_text_feature_sets``: the cached per-template-file classifiers one of which is drawn per step (:1603-1609,1662)."""

    def __init__(self, clip_model, prompt_learner: "VLPromptLearner", channel_lp: "Channel_LP",
                 zs_image_features: torch.Tensor, zs_text_features: torch.Tensor,
                 zs_text_feature_sets: Optional[Sequence[torch.Tensor]] = None, lr: float = 2e-4, total_epoch: int = 20,
                 weight_decay: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-8, moco_model=None,
                 moco_adapter: Optional["Moco_Adapter"] = None):
        self.model, self.prompt_learner, self.head = clip_model, prompt_learner, channel_lp
        self.moco_model, self.moco_adapter = moco_model, moco_adapter
        self.text_encoder = TextEncoder(clip_model)
        dev = clip_model.device
        self.zs_img = zs_image_features.to(dev, torch.float32)
        self.zs_txt = zs_text_features.to(dev, torch.float32)
        self.zs_sets = [t.to(dev, torch.float32) for t in (zs_text_feature_sets or [zs_text_features])]
        self.params: List[nn.Parameter] = [prompt_learner.ctx]
        if clip_model.visual.VPT is not None:
            clip_model.visual.VPT.requires_grad_(True)
            self.params.append(clip_model.visual.VPT)
        self.params += list(channel_lp.parameters())
        if moco_adapter is not None:
            self.params += list(moco_adapter.parameters())
        self.m = [torch.zeros_like(p.data) for p in self.params]
        self.v = [torch.zeros_like(p.data) for p in self.params]
        self.sched = CosineAnnealingLR(lr, total_epoch)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.t = 0

    def loss(self, images: torch.Tensor, target: torch.Tensor, index: torch.Tensor, template_choice: int = 0,
             raw_images: Optional[torch.Tensor] = None):
        """``images``: CLIP-normalised batch (``tfm_clip(raw)``, :1624); ``raw_images`` in [0, 1] feed the MoCo branch
        through ``tfm_moco`` (:1677) when it is enabled."""
        m = self.model
        text_features = self.text_encoder(self.prompt_learner())                  # :1626-1629
        image_features = m.encode_image(images)                                   # :1636
        with torch.no_grad():
            lp_img = m.encode_image(images)                                       # :1660-1661 (second forward)
        zs_img = self.zs_img[index.to(self.zs_img.device)]
        moco_feats = None
        if self.moco_model is not None and self.moco_adapter is not None:
            if raw_images is None:
                raise ValueError("the MoCo branch needs raw_images (in [0, 1]) next to the CLIP-normalised batch")
            moco_feats = self.moco_model(tfm_moco(raw_images.to(m.device)))           # :1677 (frozen, no grad)
        return stage2_loss(image_features, text_features, target.to(m.device), zs_img, self.zs_txt, self.head, lp_img,
                           self.zs_sets[template_choice % len(self.zs_sets)], self.moco_adapter, moco_feats)

    def step(self, images, target, index, template_choice: int = 0, raw_images: Optional[torch.Tensor] = None):
        for p in self.params:
            p.grad = None
        loss, terms, cos = self.loss(images, target, index, template_choice, raw_images)
        loss.backward()
        self.t += 1
        for p, mm, vv in zip(self.params, self.m, self.v):
            if p.grad is not None:
                ops.adamw(p.data.view(-1), p.grad.contiguous().view(-1), mm.view(-1), vv.view(-1), self.t, self.lr,
                          self.betas, self.eps, self.wd, 1.0)
        self.lr = self.sched.step()                                               # :1697
        return loss.detach(), terms, cos.detach()


@torch.no_grad()
def solve_mta(image_features: torch.Tensor, text_features: torch.Tensor) -> torch.Tensor:
    """slow_pace.py:1363-1433: returns the mode feature [1, d] (the stage-1 / ood variant returns logits)."""
    text = text_features.t().contiguous().float()
    mode, _ = ops.mta(image_features.contiguous().float().unsqueeze(0), text, want_logits=False)
    return mode
