"""Host side of the GPU view generation (csrc/views.hip): crop-box sampling and the descriptor table.

Reference pipeline (ood.py:946-958,1084-1089; lora_train_vlp.py:1203-1208; jclip/clip.py:130-144):
  view 0      preprocess = Resize(256, BICUBIC) -> CenterCrop(224) -> ImageNormalize -> ToTensor
  views 1..N  RandomResizedCrop(224, scale=(0.5, 1) or (0.2, 1)) [bilinear] -> RandomHorizontalFlip(0.5)
              -> ImageNormalize -> ToTensor          (N = 512 in the reference, 64 in cfg-4)
The box sampler follows the torchvision-style algorithm Jittor's RandomResizedCrop implements (10 attempts of
area ~ U(scale) * A, aspect ~ U(3/4, 4/3), then a centre fallback); Jittor's random stream itself cannot be
reproduced, so views are reproducible per (seed) of THIS sampler only.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
BILINEAR, BICUBIC = 0, 1
MAX_TAPS = 24


def sample_crop(width: int, height: int, scale: Tuple[float, float], ratio: Tuple[float, float],
                rng: np.random.RandomState) -> Tuple[int, int, int, int]:
    """(top, left, h, w) of one RandomResizedCrop box."""
    area = width * height
    for _ in range(10):
        target = rng.uniform(scale[0], scale[1]) * area
        aspect = rng.uniform(ratio[0], ratio[1])
        w = int(round(math.sqrt(target * aspect)))
        h = int(round(math.sqrt(target / aspect)))
        if rng.random_sample() < 0.5:
            w, h = h, w
        if 0 < w <= width and 0 < h <= height:
            top = rng.randint(0, height - h + 1)
            left = rng.randint(0, width - w + 1)
            return top, left, h, w
    w = h = min(width, height)  # fallback: central square
    return (height - h) // 2, (width - w) // 2, h, w


def centre_view_record(width: int, height: int, resize: int = 256, crop: int = 224) -> Tuple[int, ...]:
    """Resize(256): short side -> 256, long side int(256 * long / short) (clip.py:113-127); CenterCrop(224)."""
    short, long_ = (width, height) if width <= height else (height, width)
    new_long = int(resize * long_ / short)
    out_w, out_h = (resize, new_long) if width <= height else (new_long, resize)
    if short == resize:
        out_w, out_h = width, height
    win_x, win_y = int(round((out_w - crop) / 2.0)), int(round((out_h - crop) / 2.0))
    return (0, 0, height, width, 0, out_w, out_h, win_x, win_y, BICUBIC)


def view_records(width: int, height: int, n_crops: int, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), size: int = 224,
                 seed: int = 0, flip_p: float = 0.5) -> np.ndarray:
    """int32 [1 + n_crops, 10] descriptor table: the centre view then the random crops."""
    rng = np.random.RandomState(seed)
    recs = [centre_view_record(width, height, 256, size)]
    for _ in range(n_crops):
        top, left, h, w = sample_crop(width, height, scale, ratio, rng)
        flip = int(rng.random_sample() < flip_p)
        recs.append((top, left, h, w, flip, size, size, 0, 0, BILINEAR))
    recs = np.asarray(recs, dtype=np.int32)
    for r in recs:
        support = (1.0 if r[9] == BILINEAR else 2.0) * max(r[2] / r[6], r[3] / r[5], 1.0)
        if int(math.ceil(support)) * 2 + 1 > MAX_TAPS:
            raise ValueError(f"crop {tuple(r[:4])} -> {r[5]}x{r[6]} needs more than {MAX_TAPS} filter taps")
    return recs


_NORM_CACHE = {}


def _norm_constants(dev, mean, std):
    key = (str(dev), mean, std)
    hit = _NORM_CACHE.get(key)
    if hit is None:
        hit = (torch.tensor(mean, device=dev, dtype=torch.float32), torch.tensor(std, device=dev, dtype=torch.float32))
        _NORM_CACHE[key] = hit
    return hit


def make_views(image_u8: torch.Tensor, recs: np.ndarray, size: int = 224, mean: Sequence[float] = CLIP_MEAN,
               std: Sequence[float] = CLIP_STD, out: torch.Tensor = None) -> torch.Tensor:
    """image_u8: uint8 [H, W, 3] device tensor -> fp32 [n, 3, size, size] normalised views (one kernel launch), written
    into ``out`` when given (a contiguous fp32 [n, 3, size, size] slice of a caller-owned buffer)."""
    assert image_u8.is_cuda and image_u8.dtype == torch.uint8 and image_u8.dim() == 3 and image_u8.shape[2] == 3
    image_u8 = image_u8.contiguous()
    H, W = image_u8.shape[:2]
    dev = image_u8.device
    # no host-blocking copy here: a pageable upload waits for everything queued on the current stream, which stalls a
    # caller that generates views on a side stream under another stream's GEMMs (ood.score_stream)
    r = torch.from_numpy(np.ascontiguousarray(recs, dtype=np.int32)).pin_memory().to(dev, non_blocking=True)
    n = r.shape[0]
    if out is None:
        out = torch.empty(n, 3, size, size, device=dev, dtype=torch.float32)
    elif tuple(out.shape) != (n, 3, size, size) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
        raise ValueError(f"make_views: out must be a contiguous fp32 [{n}, 3, {size}, {size}] tensor on {dev}")
    m, s = _norm_constants(dev, tuple(float(v) for v in mean), tuple(float(v) for v in std))
    check(_lib.load().clipfs_tta_views(image_u8.data_ptr(), H, W, r.data_ptr(), n, size, m.data_ptr(), s.data_ptr(),
                                       out.data_ptr(), torch.cuda.current_stream().cuda_stream), "tta_views")
    return out
