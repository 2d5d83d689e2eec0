"""CLIP model objects with the attribute surface of the reference's ``jclip/model.py`` (``CLIP``,
``VisionTransformer``, ``Transformer``, ``ResidualAttentionBlock``, ``build_model``) and, with
``design_details``, of ``jclip/model1.py`` (shallow VPT tokens, ``VisionTransformer.VPT``).

These classes only HOLD parameters (device tensors, OpenAI-CLIP state-dict names) and expose the
replaceable ``resblocks[i].attn`` that ``apply_lora`` swaps.  All arithmetic runs in the HIP engine:
``encode_image`` / ``encode_text`` hand the whole tower to ``clipfs.engine`` (one C call per pass),
differentiable through ``torch.autograd`` with respect to the LoRA / prompt parameters only -- the
backbone is frozen (dgrad only), which is the reference's training regime
(mark_only_lora_as_trainable, lora_train_vlp.py:143-160)."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from .mha import MultiheadAttention


def _param(t: torch.Tensor) -> nn.Parameter:
    return nn.Parameter(t.contiguous(), requires_grad=False)


class LayerNorm(nn.Module):
    def __init__(self, weight: torch.Tensor, bias: torch.Tensor):
        super().__init__()
        self.weight = _param(weight)
        self.bias = _param(bias)
        self.eps = 1e-5


class Linear(nn.Module):
    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        super().__init__()
        self.weight = _param(weight)
        self.bias = _param(bias) if bias is not None else None
        self.out_features, self.in_features = weight.shape


class MLP(nn.Module):
    def __init__(self, sd: Dict[str, torch.Tensor], p: str):
        super().__init__()
        self.c_fc = Linear(sd[p + "c_fc.weight"], sd[p + "c_fc.bias"])
        self.c_proj = Linear(sd[p + "c_proj.weight"], sd[p + "c_proj.bias"])


class ResidualAttentionBlock(nn.Module):
    """jclip/model.py:42-62 (parameter holder; see module docstring)."""

    def __init__(self, sd: Dict[str, torch.Tensor], p: str, d_model: int, n_head: int, causal: bool):
        super().__init__()
        self.attn = MultiheadAttention(d_model, n_head, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                                       sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        self.ln_1 = LayerNorm(sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        self.mlp = MLP(sd, p + "mlp.")
        self.ln_2 = LayerNorm(sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        self.causal = causal


class Transformer(nn.Module):
    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, width: int, layers: int, heads: int, causal: bool):
        super().__init__()
        self.width = width
        self.layers = layers
        self.heads = heads
        self.causal = causal
        self.resblocks = nn.Sequential(*[
            ResidualAttentionBlock(sd, f"{prefix}.resblocks.{i}.", width, heads, causal) for i in range(layers)])


class _Conv1(nn.Module):
    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.weight = _param(weight)


class VisionTransformer(nn.Module):
    """jclip/model.py:80-126; with ``n_vpt > 0`` the shallow-VPT variant of jclip/model1.py:160-207."""

    def __init__(self, sd: Dict[str, torch.Tensor], input_resolution: int, patch_size: int, width: int, layers: int,
                 heads: int, output_dim: int, n_vpt: int = 0):
        super().__init__()
        self.input_resolution = input_resolution
        self.patch_size = patch_size
        self.output_dim = output_dim
        self.width = width
        self.conv1 = _Conv1(sd["visual.conv1.weight"])
        self.class_embedding = _param(sd["visual.class_embedding"])
        self.positional_embedding = _param(sd["visual.positional_embedding"])
        self.ln_pre = LayerNorm(sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
        self.transformer = Transformer(sd, "visual.transformer", width, layers, heads, causal=False)
        self.ln_post = LayerNorm(sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
        self.proj = _param(sd["visual.proj"])
        if n_vpt > 0:
            if "visual.VPT" in sd:
                vpt = sd["visual.VPT"]
            else:  # normal_(ctx_vectors, std=0.02), model1.py:161-163
                g = torch.Generator().manual_seed(0)
                vpt = (torch.randn(n_vpt, width, generator=g) * 0.02).to(self.proj.device)
            self.VPT = nn.Parameter(vpt.contiguous())
        else:
            self.VPT = None

    @property
    def tokens(self) -> int:
        return (self.input_resolution // self.patch_size) ** 2 + 1 + (0 if self.VPT is None else self.VPT.shape[0])


class _Embedding(nn.Module):
    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.weight = _param(weight)

    @torch.no_grad()
    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        """Plain lookup (used by the prompt learner to initialise ctx, slow_pace.py:124-131)."""
        return self.weight.data[ids.to(self.weight.device).long()]

    execute = forward


class CLIP(nn.Module):
    """jclip/model.py:129-232."""

    def __init__(self, sd: Dict[str, torch.Tensor], embed_dim: int, image_resolution: int, vision_layers: int,
                 vision_width: int, vision_patch_size: int, context_length: int, vocab_size: int,
                 transformer_width: int, transformer_heads: int, transformer_layers: int, design_details=None):
        super().__init__()
        self.context_length = context_length
        self.vocab_size = vocab_size
        self.embed_dim = embed_dim
        n_vpt = int(design_details["vision_ctx"]) if design_details else 0
        self.visual = VisionTransformer(sd, image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim, n_vpt)
        self.transformer = Transformer(sd, "transformer", transformer_width, transformer_layers, transformer_heads,
                                       causal=True)
        self.token_embedding = _Embedding(sd["token_embedding.weight"])
        self.positional_embedding = _param(sd["positional_embedding"])
        self.ln_final = LayerNorm(sd["ln_final.weight"], sd["ln_final.bias"])
        self.text_projection = _param(sd["text_projection"])
        self.logit_scale = _param(sd["logit_scale"].reshape(()))
        self._engine = None

    # -- reference surface ---------------------------------------------------------------------
    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    @property
    def device(self):
        return self.visual.conv1.weight.device

    # -- Jittor Module.save / Module.load (slow_pace.py:1711, test.py:1820) -----------------------------------------
    def save(self, path: str) -> None:
        """``clip_model.save('test_pkl/clip_model.pkl')``: every parameter under its dotted name (with adapters:
        ``...attn.q_proj.weight`` / ``.bias`` / ``.w_lora_A`` / ``.w_lora_B`` ..., as the reference's module tree)."""
        from clipfs import module_io
        module_io.save_module(self, path)

    def load(self, path: str) -> None:
        from clipfs import module_io
        module_io.load_module(self, path)
        self.invalidate_engine()  # transposed / 16-bit weight copies of the engine are stale now

    def invalidate_engine(self):
        """Called when the module tree changes (apply_lora, FlatTrainables) so pointers are re-collected.  The
        engine's user-visible settings (precision mode, text trimming) carry over to the rebuilt engine."""
        if self._engine is not None:
            self._engine_opts = (self._engine.precision, self._engine.trim_text, self._engine.sparse_backward)
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            from clipfs.engine import Engine
            self._engine = Engine(self)
            opts = getattr(self, "_engine_opts", None)
            if opts is not None:
                self._engine.precision, self._engine.trim_text, self._engine.sparse_backward = opts
        return self._engine

    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        from clipfs.engine import encode_image
        return encode_image(self, image)

    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        from clipfs.engine import encode_text
        return encode_text(self, text)

    def forward(self, image, text):
        """CLIP.execute, jclip/model.py:217-232."""
        from clipfs.engine import clip_logits
        return clip_logits(self, image, text)

    execute = forward

    def cuda(self, device=None):  # tensors are created on the target device by build_model
        return self


def _infer(sd: Dict[str, torch.Tensor]):
    """Hyper-parameters from tensor shapes, jclip/model.py:235-274."""
    if "visual.proj" not in sd:
        raise NotImplementedError("only ViT backbones are on the accelerated path (ModifiedResNet: SURVEY.md section 2 row 16)")
    vision_width = sd["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = sd["visual.conv1.weight"].shape[-1]
    grid_size = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    image_resolution = vision_patch_size * grid_size
    embed_dim = sd["text_projection"].shape[1]
    context_length = sd["positional_embedding"].shape[0]
    vocab_size = sd["token_embedding.weight"].shape[0]
    transformer_width = sd["ln_final.weight"].shape[0]
    transformer_heads = transformer_width // 64
    transformer_layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")})
    return dict(embed_dim=embed_dim, image_resolution=image_resolution, vision_layers=vision_layers,
                vision_width=vision_width, vision_patch_size=vision_patch_size, context_length=context_length,
                vocab_size=vocab_size, transformer_width=transformer_width, transformer_heads=transformer_heads,
                transformer_layers=transformer_layers)


def build_model(state_dict: dict, design_details=None, device=None) -> CLIP:
    """jclip/model.py:235-285 (and model1.py:322-374 with ``design_details``).  ``state_dict`` values may be
    numpy arrays (a Jittor-saved pkl read by clipfs.safe_pkl) or torch tensors; they are moved to
    ``device`` (default cuda:0) as fp32.  The engine has no CPU path: a missing GPU raises."""
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("jclip needs an MI355X: the HIP engine has no CPU fallback")
        device = torch.device("cuda", torch.cuda.current_device())
    sd = {}
    for k, v in state_dict.items():
        if k in ("input_resolution", "context_length", "vocab_size"):
            continue
        t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else torch.as_tensor(v)
        sd[k] = t.detach().to(device=device, dtype=torch.float32).contiguous()
    model = CLIP(sd, design_details=design_details, **_infer(sd))
    return model.eval()
