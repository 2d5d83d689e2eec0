"""``jclip.clip``: ``available_models`` / ``load`` / ``tokenize`` with the reference's signatures
(jclip/clip.py:16,165-214) on the HIP engine.

* ``load(name)``: ``name`` is a local checkpoint path, as in every call site of the reference
  (lora_train_vlp.py:1163).  Accepted formats: a Jittor-saved ``.pkl`` state dict (read with the inert
  reader ``clipfs.safe_pkl`` -- nothing in the file is executed), a ``.npz`` of arrays, or a torch
  ``.pt`` state dict (``weights_only=True``).  A registry name (``"ViT-B/32"``) would need a download
  (clip.py:171-174): there is no network, so it raises the reference's RuntimeError.
* returns the reference's 5-tuple ``(model, tfm_nonorm, tfm_norm, tfm_train_nonorm, tfm_train_norm)``
  (clip.py:187); the transforms are PIL -> float32 CHW tensors (Resize 256 bicubic, CenterCrop 224,
  optional horizontal flip / CLIP mean-std normalisation, clip.py:130-163).
* ``tokenize``: int64 [N, 77] on the host, SOT + BPE + EOT, zero padded (clip.py:190-214).
"""
from __future__ import annotations

import os
import random
from typing import List, Union

import numpy as np
import torch

from .model import build_model
from .simple_tokenizer import SimpleTokenizer as _Tokenizer

__all__ = ["available_models", "load", "tokenize"]

_MODELS = ["RN50", "RN101", "RN50x4", "RN50x16", "RN50x64", "ViT-B/32", "ViT-B/16", "ViT-L/14", "ViT-L/14@336px"]
_tokenizer = None

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _tok() -> _Tokenizer:
    global _tokenizer
    if _tokenizer is None:
        _tokenizer = _Tokenizer()
    return _tokenizer


def available_models() -> List[str]:
    """Returns the names of available CLIP models"""
    return list(_MODELS)


# ---- transforms (clip.py:102-163) ---------------------------------------------------------------

class _Transform:
    def __init__(self, n_px: int, normalize: bool, flip: bool):
        self.n_px, self.normalize, self.flip = n_px, normalize, flip

    def __call__(self, img):
        from PIL import Image
        if not isinstance(img, Image.Image):
            img = Image.fromarray(np.asarray(img))
        img = img.convert("RGB")
        if self.flip and random.random() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        w, h = img.size
        short, long_ = (w, h) if w <= h else (h, w)
        if short != 256:  # Resize(256): short side to 256, long side int(256 * long / short)  (clip.py:113-127)
            new_long = int(256 * long_ / short)
            new_w, new_h = (256, new_long) if w <= h else (new_long, 256)
            img = img.resize((new_w, new_h), Image.BICUBIC)
        w, h = img.size
        c = self.n_px
        left, top = int(round((w - c) / 2.0)), int(round((h - c) / 2.0))
        img = img.crop((left, top, left + c, top + c))
        x = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0
        if self.normalize:
            x = (x - np.asarray(CLIP_MEAN, np.float32)[:, None, None]) / np.asarray(CLIP_STD, np.float32)[:, None, None]
        return torch.from_numpy(np.ascontiguousarray(x))


def _transform1(n_px):
    return _Transform(n_px, normalize=False, flip=False)


def _transform2(n_px):
    return _Transform(n_px, normalize=True, flip=False)


def tfm_train_base(n_px):
    return _Transform(n_px, normalize=False, flip=True)


def tfm_train_base1(n_px):
    return _Transform(n_px, normalize=True, flip=True)


# ---- checkpoint reading -------------------------------------------------------------------------

def read_state_dict(path: str) -> dict:
    from clipfs import safe_pkl
    with open(path, "rb") as f:
        head = f.read(4)
    if head[:2] == b"PK":  # zip container: .npz (members *.npy) or a torch zip checkpoint (archive/data.pkl ...)
        import zipfile
        with zipfile.ZipFile(path) as zf:
            names = zf.namelist()
        if names and all(n.endswith(".npy") for n in names):
            with np.load(path, allow_pickle=False) as z:
                return {k: z[k] for k in z.files}
        sd = torch.load(path, map_location="cpu", weights_only=True)
        return sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    return safe_pkl.load(path)


def _load(name, design_details=None, mode="vit", device=None):
    if name in _MODELS:
        raise RuntimeError(f"Model {name} would have to be downloaded (jclip/clip.py:171-174); no network here. "
                           f"Pass the path of a local checkpoint instead.")
    if not os.path.isfile(name):
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    if mode != "vit":
        raise NotImplementedError("ModifiedResNet backbones are outside the accelerated path (SURVEY.md section 2 row 16)")
    model = build_model(read_state_dict(name), design_details=design_details, device=device)
    r = model.visual.input_resolution
    return model, _transform1(r), _transform2(r), tfm_train_base(r), tfm_train_base1(r)


def load(name, download_root=None, mode="vit", device=None):
    """jclip/clip.py:170-187."""
    return _load(name, None, mode, device)


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
    """jclip/clip.py:190-214."""
    if isinstance(texts, str):
        texts = [texts]
    tk = _tok()
    sot, eot = tk.encoder["<|startoftext|>"], tk.encoder["<|endoftext|>"]
    result = torch.zeros(len(texts), context_length, dtype=torch.int64)
    for i, text in enumerate(texts):
        tokens = [sot] + tk.encode(text) + [eot]
        if len(tokens) > context_length:
            if truncate:
                tokens = tokens[:context_length]
                tokens[-1] = eot
            else:
                raise RuntimeError(f"Input {texts[i]} is too long for context length {context_length}")
        result[i, :len(tokens)] = torch.tensor(tokens, dtype=torch.int64)
    return result
