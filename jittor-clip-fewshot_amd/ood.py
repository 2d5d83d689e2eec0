"""OOD split scoring of the reference's ``ood.py`` on the HIP engine.

    cls_acc       ood.py:638-652   accuracy of the base/new decision
    split_ood     ood.py:857-883   per image: 1 centre view + N crops -> encode -> normalise -> MTA ->
                                   argmax <= 372 -> base list, else new list
The file writing / dataset loop of the reference stays on the host; here are the batched GPU parts.
Quirk kept: the split uses ``<= 372`` (ood.py:880) although base classes are 0..373 in classes.txt, and
cls_acc uses ``< 373`` -- the same boundary.
"""
from __future__ import annotations

from typing import Tuple

import torch

from clipfs import ops

BASE_BOUNDARY = 372


def cls_acc(output: torch.Tensor, target: torch.Tensor, topk: int = 1) -> float:
    pred = ops.topk(output.contiguous().float(), topk).long().t()
    t = target.to(pred.device).view(1, -1)
    correct = ((pred < 373) & (t < 373)) | ((pred >= 373) & (t >= 373))
    return 100.0 * float(correct[:topk].reshape(-1).float().sum().item()) / target.shape[0]


@torch.no_grad()
def mta_scores(clip_model, views: torch.Tensor, text_features_cd: torch.Tensor, want_mode: bool = False):
    """views [n_img, V, 3, R, R] (view 0 = centre crop), text_features_cd [C, d] unit rows.
    One image-tower pass over all n_img*V views, one L2-normalise, one MTA launch (a workgroup per image).
    Returns (mta logits [n_img, C] = 100 * mode . text, mode [n_img, d] or None)."""
    n_img, V = views.shape[:2]
    feats = clip_model.encode_image(views.reshape(n_img * V, *views.shape[2:]))
    feats = ops.l2norm_fwd(feats.contiguous())
    mode, logits = ops.mta(feats.reshape(n_img, V, -1), text_features_cd.contiguous().float(), want_mode=want_mode)
    return logits, mode


@torch.no_grad()
def split_ood(clip_model, views: torch.Tensor, text_features_cd: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """ood.py:873-883 for a batch of images: returns (is_base [n_img] bool, pred [n_img] int64)."""
    logits, _ = mta_scores(clip_model, views, text_features_cd)
    pred = ops.topk(logits, 1).long().squeeze(1)
    return pred <= BASE_BOUNDARY, pred
