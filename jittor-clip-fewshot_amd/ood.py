"""OOD split scoring of the reference's ``ood.py`` on the HIP engine.

    cls_acc       ood.py:638-652   accuracy of the base/new decision
    split_ood     ood.py:857-883   per image: 1 centre view + N crops -> encode -> normalise -> MTA ->
                                   argmax <= 372 -> base list, else new list
    score_stream  ood.py:946-958   the image loop itself as a three-stream pipeline (views | tower | MTA)
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


@torch.no_grad()
def score_views(clip_model, views: torch.Tensor, text_features_cd: torch.Tensor):
    """One image-tower pass over n_img * V views -> (top-5 labels [n_img, 5] int32, is_base [n_img] bool, MTA logits):
    the scoring of ood.py:867-883 and the label list of test.py:1738 from the same features."""
    logits, _ = mta_scores(clip_model, views, text_features_cd)
    top5 = ops.topk(logits, 5)
    return top5, top5[:, 0].long() <= BASE_BOUNDARY, logits


@torch.no_grad()
def score_stream(clip_model, sources, text_features_cd: torch.Tensor, n_crops: int = 64, images_per_pass: int = 8,
                 seed: int = 0, scale=(0.5, 1.0)):
    """cfg-4 over a LIST of source images (PIL / uint8 arrays / uint8 tensors), ``images_per_pass`` images per tower pass,
    as a three-stage pipeline: while the image tower runs group g on the caller's stream, the views of group g + 1
    (csrc/views.hip: small, memory-bound kernels that fit beside the GEMMs) and the MTA + top-5 of group g - 1 (one
    workgroup per image: 8 of 256 CUs for ~3 ms) run on the process's high-priority side stream.  The
    reference overlaps the same stages with DataLoader workers (ood.py:946-958); per group the kernels, seeds (image i
    draws its crops from ``seed + i``) and results are those of ``score_views`` on ``tta.make_tta_views`` -- only the
    order in which the GPU sees them changes.  Returns (top5 [n, 5] int32, is_base [n] bool, logits [n, C])."""
    import tta
    text = text_features_cd.contiguous().float()
    dev = text.device
    n = len(sources)
    if n == 0:
        raise ValueError("score_stream: no source images")
    main = torch.cuda.current_stream(dev)
    # ONE side stream for both overlapped stages, and the same one the trainer uses (clipfs/streams.py: every further
    # stream is another draw of HIP's stream -> hardware-queue mapping, and an unlucky one serialises somebody's overlap
    # without any error).  Order on it per group: views of g + 1, then MTA of g (which waits for the tower pass of g).
    from clipfs import streams
    s_views = s_mta = streams.side_stream(dev)
    groups = [(lo, min(lo + images_per_pass, n)) for lo in range(0, n, images_per_pass)]
    V, size = 1 + n_crops, 224
    # two view buffers owned by this call, written in place by the view kernels (no per-group allocation: a fresh block on
    # a side stream cannot be recycled before the consumer's stream has passed it, so every group would go to hipMalloc)
    bufs = [torch.empty(min(images_per_pass, n), V, 3, size, size, device=dev, dtype=torch.float32) for _ in range(min(2, len(groups)))]
    consumed = [None, None]  # event: the tower pass that read the buffer last has finished

    def generate(g):
        lo, hi = groups[g]
        buf = bufs[g % 2][:hi - lo]
        with torch.cuda.stream(s_views):
            if consumed[g % 2] is not None:
                s_views.wait_event(consumed[g % 2])
            for i in range(lo, hi):
                tta.make_tta_views(sources[i], n_crops, scale=scale, seed=seed + i, size=size, device=dev, out=buf[i - lo])
            ev = torch.cuda.Event()
            ev.record(s_views)
        return buf, ev

    s_views.wait_stream(main)  # the sources may have been uploaded on the caller's stream
    nxt = generate(0)
    top5s, logit_list = [], []
    for g in range(len(groups)):
        views, ev = nxt
        main.wait_event(ev)
        n_img = views.shape[0]
        feats = clip_model.encode_image(views.reshape(n_img * V, 3, size, size))
        done = torch.cuda.Event()
        done.record(main)
        consumed[g % 2] = done
        if g + 1 < len(groups):
            nxt = generate(g + 1)  # on the side stream: runs under group g's tower pass
        feats = ops.l2norm_fwd(feats.contiguous())
        evf = torch.cuda.Event()
        evf.record(main)
        with torch.cuda.stream(s_mta):
            s_mta.wait_event(evf)
            feats.record_stream(s_mta)
            _, logits = ops.mta(feats.reshape(n_img, V, -1), text)
            top5s.append(ops.topk(logits, 5))
            logit_list.append(logits)
    main.wait_stream(s_mta)
    main.wait_stream(s_views)
    for t in top5s + logit_list:
        t.record_stream(main)
    top5 = torch.cat(top5s)
    return top5, top5[:, 0].long() <= BASE_BOUNDARY, torch.cat(logit_list)


@torch.no_grad()
def score_images_sharded(score_fn, n_images: int, device, group=None):
    """cfg-4 on N GPUs: source images are independent units, rank r scores images
    [lo, hi) -- every image's views stay on one GPU, no data-path collective -- and the integer results are
    assembled on every rank with one exchange.  ``score_fn(lo, hi)`` returns (top5 [hi - lo, 5], is_base [hi - lo])
    for the local shard (e.g. ``lambda lo, hi: score_views(model, views[lo:hi], text)[:2]``).
    Returns (top5 [n, 5] int32, is_base [n] bool) identical on all ranks."""
    from clipfs import dist as D
    rank, world = D.world_info(group)
    lo, hi = D.shard_bounds(n_images, rank, world)
    local = None
    if hi > lo:
        top5, is_base = score_fn(lo, hi)
        local = torch.cat([top5.to(torch.int32).reshape(hi - lo, 5), is_base.to(torch.int32).reshape(hi - lo, 1)], dim=1)
    table = D.allgather_int_rows(local, lo, hi, n_images, 6, device, group)
    return table[:, :5].contiguous(), table[:, 5] != 0
