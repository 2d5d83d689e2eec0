"""Crop-batch scoring and top-5 fusion of the reference's ``test.py`` (evaluate_base, :1675-1747) on the
HIP engine: three MTA passes per image (LoRA+prompt text, LoRA+hand-written text, zero-shot model),
logit fusion, top-5 labels.  The txt-file merging of the reference stays on the host (out of scope)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from clipfs import ops


def make_tta_views(image, n_crops: int = 512, scale=(0.5, 1.0), seed: int = 0, size: int = 224, device=None, out=None):
    """The 1 + n_crops views of one image as a device tensor [1 + n_crops, 3, size, size] (view 0 = the centre
    preprocess, the rest RandomResizedCrop(scale) + flip), generated ON THE GPU from the uint8 image
    (csrc/views.hip, pixel-exact with PIL).  Replaces the reference's CPU worker loop ood.py:946-958 /
    lora_train_vlp.py:1065-1077.  ``image``: PIL image, uint8 numpy [H, W, 3] or uint8 tensor."""
    import numpy as np
    from clipfs import views
    if hasattr(image, "convert"):
        image = np.asarray(image.convert("RGB"))
    if isinstance(image, np.ndarray):
        image = torch.from_numpy(np.ascontiguousarray(image))
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    image = image.to(device)
    H, W = image.shape[:2]
    recs = views.view_records(W, H, n_crops, scale=scale, seed=seed, size=size)
    return views.make_views(image, recs, size, out=out)


@torch.no_grad()
def fuse_top5(cos: torch.Tensor, cos1: torch.Tensor, cos3: torch.Tensor,
              head_logits: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """test.py:1729-1742.  cos = 100*mode_hand.text, cos1 = 100*mode_pt.text1, cos3 = 100*mode_zs.text_zs.
    cos2 = (cos + cos1)/2, cos4 = (cos2 + cos3)/2, cos5 = cos4 + 0.5*head; the written top-5 comes from cos1."""
    cos2 = (cos + cos1) / 2
    cos4 = (cos2 + cos3) / 2
    out = {"cos2": cos2, "cos4": cos4, "top5": ops.topk(cos1.contiguous().float(), 5)}
    if head_logits is not None:
        out["cos5"] = cos4 + 0.5 * head_logits
    return out


@torch.no_grad()
def evaluate_views(model, model_zs, views, text_pt, text_hand, text_zs, channel_lp=None):
    """One batch of test.py:evaluate_base: views [n_img, V, 3, R, R]; text_* are [C, d] unit rows.
    Returns the fusion dict (per image rows)."""
    from slow_pace import logit_normalize
    n_img, V = views.shape[:2]
    flat = views.reshape(n_img * V, *views.shape[2:])
    f = ops.l2norm_fwd(model.encode_image(flat).contiguous()).reshape(n_img, V, -1)
    mode_pt, cos1 = ops.mta(f, text_pt.contiguous().float())
    mode_hand, cos = ops.mta(f, text_hand.contiguous().float())
    fz = ops.l2norm_fwd(model_zs.encode_image(flat).contiguous()).reshape(n_img, V, -1)
    mode_zs, cos3 = ops.mta(fz, text_zs.contiguous().float())
    head = None
    if channel_lp is not None:
        l1 = logit_normalize(channel_lp((mode_pt + mode_hand) / 2))
        l2 = logit_normalize(channel_lp(mode_zs))
        head = logit_normalize((l1 + l2) / 2)
    return fuse_top5(cos, cos1, cos3, head)
