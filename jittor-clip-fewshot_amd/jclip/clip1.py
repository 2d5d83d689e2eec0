"""``jclip.clip1``: ``load_vlp`` -- the shallow-VPT model variant (reference jclip/clip1.py:189-213,
jclip/model1.py): same loader as ``jclip.clip.load`` plus 4 learnable visual prompt tokens appended
after the patch tokens (``model.visual.VPT``, model1.py:160-164,192-194)."""
from __future__ import annotations

from .clip import _load, available_models, load, tokenize  # noqa: F401

__all__ = ["available_models", "load", "load_vlp", "tokenize"]

DESIGN_DETAILS = {"trainer": "IVLP", "vision_depth": 3, "language_depth": 3, "vision_ctx": 4, "language_ctx": 4}


def load_vlp(name, download_root=None, mode="vit", device=None):
    """jclip/clip1.py:189-213 (deep prompts are coded but disabled in the reference: prompts_needed=0)."""
    return _load(name, dict(DESIGN_DETAILS), mode, device)
