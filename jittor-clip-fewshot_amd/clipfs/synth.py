"""Synthetic CLIP weights / inputs (there is no network: the pretrained
``ViT-B-32.pkl`` is absent).  State-dict keys use the OpenAI CLIP
naming the reference's ``build_model`` consumes (jclip/model.py:235-285).
Statistics follow CLIP.initialize_parameters (jclip/model.py:172-187) and
VisionTransformer.__init__ (:93-102); spec in SURVEY.md section 8d.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch


@dataclass(frozen=True)
class ClipConfig:
    name: str
    embed_dim: int
    image_resolution: int
    vision_layers: int
    vision_width: int
    vision_patch_size: int
    context_length: int
    vocab_size: int
    transformer_width: int
    transformer_layers: int

    @property
    def vision_heads(self) -> int:
        return self.vision_width // 64

    @property
    def transformer_heads(self) -> int:
        return self.transformer_width // 64

    @property
    def vision_tokens(self) -> int:
        return (self.image_resolution // self.vision_patch_size) ** 2 + 1


VIT_B32 = ClipConfig("ViT-B/32", 512, 224, 12, 768, 32, 77, 49408, 512, 12)
VIT_L14 = ClipConfig("ViT-L/14", 768, 224, 24, 1024, 14, 77, 49408, 768, 12)
# small shapes for CPU-speed parity tests (same code paths: head_dim 64, ragged M)
TINY = ClipConfig("tiny", 64, 64, 2, 128, 32, 16, 512, 64, 2)
SMALL = ClipConfig("small", 128, 96, 3, 192, 32, 24, 1024, 128, 3)


def synth_state_dict(cfg: ClipConfig, seed: int = 1234, perturb: bool = False,
                     dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """CLIP-init random weights.  ``perturb=True`` additionally randomises biases
    and LayerNorm affine parameters (zero / one at init) so that parity tests
    exercise those terms."""
    g = torch.Generator().manual_seed(seed)

    def normal(*shape, std=1.0):
        return (torch.randn(*shape, generator=g, dtype=torch.float64) * std).to(dtype)

    sd: Dict[str, torch.Tensor] = {}
    vw, ps = cfg.vision_width, cfg.vision_patch_size
    sd["visual.conv1.weight"] = normal(vw, 3, ps, ps, std=(3 * ps * ps) ** -0.5)
    sd["visual.class_embedding"] = normal(vw, std=vw ** -0.5)
    sd["visual.positional_embedding"] = normal(cfg.vision_tokens, vw, std=vw ** -0.5)
    sd["visual.proj"] = normal(vw, cfg.embed_dim, std=vw ** -0.5)

    def ln(prefix, width):
        if perturb:
            sd[prefix + ".weight"] = (1.0 + normal(width, std=0.1)).to(dtype)
            sd[prefix + ".bias"] = normal(width, std=0.1)
        else:
            sd[prefix + ".weight"] = torch.ones(width, dtype=dtype)
            sd[prefix + ".bias"] = torch.zeros(width, dtype=dtype)

    def bias(n):
        return normal(n, std=0.02) if perturb else torch.zeros(n, dtype=dtype)

    def blocks(prefix, width, layers):
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        attn_std = width ** -0.5
        fc_std = (2 * width) ** -0.5
        for i in range(layers):
            p = f"{prefix}.resblocks.{i}."
            ln(p + "ln_1", width)
            sd[p + "attn.in_proj_weight"] = normal(3 * width, width, std=attn_std)
            sd[p + "attn.in_proj_bias"] = bias(3 * width)
            sd[p + "attn.out_proj.weight"] = normal(width, width, std=proj_std)
            sd[p + "attn.out_proj.bias"] = bias(width)
            ln(p + "ln_2", width)
            sd[p + "mlp.c_fc.weight"] = normal(4 * width, width, std=fc_std)
            sd[p + "mlp.c_fc.bias"] = bias(4 * width)
            sd[p + "mlp.c_proj.weight"] = normal(width, 4 * width, std=proj_std)
            sd[p + "mlp.c_proj.bias"] = bias(width)

    ln("visual.ln_pre", vw)
    blocks("visual.transformer", vw, cfg.vision_layers)
    ln("visual.ln_post", vw)

    tw = cfg.transformer_width
    sd["token_embedding.weight"] = normal(cfg.vocab_size, tw, std=0.02)
    sd["positional_embedding"] = normal(cfg.context_length, tw, std=0.01)
    blocks("transformer", tw, cfg.transformer_layers)
    ln("ln_final", tw)
    sd["text_projection"] = normal(tw, cfg.embed_dim, std=tw ** -0.5)
    sd["logit_scale"] = torch.tensor(float(np.log(1 / 0.07)), dtype=dtype)
    return sd


def synth_images(batch: int, resolution: int = 224, seed: int = 0) -> torch.Tensor:
    """fp32 [B,3,R,R] ~ N(0,1) (post-normalisation statistics)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, resolution, resolution, generator=g, dtype=torch.float32)


def synth_captions(n: int, context_length: int = 77, vocab_size: int = 49408, seed: int = 1,
                   min_len: int = 6, max_len: int = 40) -> torch.Tensor:
    """int64 [n, ctx]: SOT, len~U[min,max] ids ~U[1, vocab-3], EOT, zeros.  SOT/EOT
    are the two highest ids (49406/49407 for the real vocabulary)."""
    rng = np.random.RandomState(seed)
    sot, eot = vocab_size - 2, vocab_size - 1
    max_len = min(max_len, context_length - 2)
    min_len = min(min_len, max_len)
    out = np.zeros((n, context_length), dtype=np.int64)
    for i in range(n):
        ln_ = rng.randint(min_len, max_len + 1)
        out[i, 0] = sot
        out[i, 1:1 + ln_] = rng.randint(1, vocab_size - 2, size=ln_)
        out[i, 1 + ln_] = eot
    return torch.from_numpy(out)


def synth_labels(n: int, n_classes: int = 374, seed: int = 2) -> torch.Tensor:
    rng = np.random.RandomState(seed)
    return torch.from_numpy(rng.randint(0, n_classes, size=n).astype(np.int64))


def synth_lora(cfg: ClipConfig, r: int, seed: int = 5, params=("q", "k", "v"),
               text_blocks=None, vision_blocks=None) -> Dict[str, dict]:
    """LoRA weights in the save_lora schema (lora_train_vlp.py:551-593):
    A ~ U(+-1/sqrt(in)) (kaiming_uniform a=sqrt(5), :212), B ~ N(0, 0.02^2) (the
    reference initialises B to zero, :213; non-zero here so the adapter path is
    exercised).  Layer order: text blocks then vision blocks."""
    g = torch.Generator().manual_seed(seed)
    names = {"q": "q_proj", "k": "k_proj", "v": "v_proj", "o": "proj"}
    text_blocks = range(cfg.transformer_layers) if text_blocks is None else text_blocks
    vision_blocks = range(cfg.vision_layers) if vision_blocks is None else vision_blocks
    weights = {}
    i = 0
    for width, blks in ((cfg.transformer_width, text_blocks), (cfg.vision_width, vision_blocks)):
        for _ in blks:
            layer = {}
            for p in params:
                a = ((torch.rand(r, width, generator=g, dtype=torch.float64) * 2 - 1) / width ** 0.5)
                b = torch.randn(width, r, generator=g, dtype=torch.float64) * 0.02
                layer[names[p]] = {"w_lora_A": a.float().numpy(), "w_lora_B": b.float().numpy()}
            weights[f"layer_{i}"] = layer
            i += 1
    return weights
