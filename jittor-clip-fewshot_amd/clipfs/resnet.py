"""Frozen ResNet-50 feature extractor of the MoCo-v3 auxiliary branch (reference: ``load_moco`` slow_pace.py:1237-1271,
which builds ``jittor.models.resnet.resnet50`` -- torchvision's ResNet-50 v1.5, stride on the 3x3 convolution -- loads the
MoCo-v3 ``base_encoder.*`` weights and replaces ``fc`` by Identity; used forward-only at :1158,1677,1013,1099).

Execution on the HIP engine: activations NHWC; every convolution is one fp32-MFMA GEMM (csrc/gemm.hip) -- a 1x1 /
stride-1 convolution directly on the activation tensor, the others on an im2col matrix (csrc/resnet.hip) -- with the
BatchNorm of inference mode folded into weight and bias ONCE at load time and bias + residual + ReLU in the GEMM epilogue.

Flagged deviation: BatchNorm always uses the RUNNING statistics (a frozen feature extractor; ``moco_model.eval()`` is what
test.py:1828 and evaluate_lora :950 set).  The reference's training script never calls ``moco_model.eval()`` before its
first epoch, so there Jittor's default training flag makes the first epoch (and pre_load_features_moco) use batch
statistics -- an accident of the script, not reproduced.  Pretrained weights (r-50-1000ep.pkl) are not available
offline: parity is the oracle's (oracle/clip_oracle.py: resnet50_forward) on synthetic weights, i.e. unpinned against
Jittor like the rest of the path.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import _lib, ops
from ._lib import check

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
BN_EPS = 1e-5


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Conv:
    """One convolution + folded BatchNorm as a GEMM operand: weight [Cout, Kp] in (ky, kx, cin) column order."""

    def __init__(self, sd: Dict[str, torch.Tensor], conv: str, bn: str, stride: int, pad: int, device):
        w = sd[conv + ".weight"].to(torch.float64)
        cout, cin, kh, kw = w.shape
        g, b = sd[bn + ".weight"].double(), sd[bn + ".bias"].double()
        mu, var = sd[bn + ".running_mean"].double(), sd[bn + ".running_var"].double()
        scale = g / torch.sqrt(var + BN_EPS)
        w = (w * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(cout, kh * kw * cin)
        k = kh * kw * cin
        self.kp = (k + 3) // 4 * 4
        if self.kp != k:
            w = torch.cat([w, torch.zeros(cout, self.kp - k, dtype=w.dtype)], dim=1)
        self.weight = w.float().contiguous().to(device)
        self.bias = (b - mu * scale).float().contiguous().to(device)
        self.cin, self.cout, self.kh, self.kw, self.stride, self.pad = cin, cout, kh, kw, stride, pad

    def __call__(self, x: torch.Tensor, shape: Tuple[int, int, int], relu: bool, residual=None):
        """x [N*H*W, cin] (NHWC rows), shape = (N, H, W) -> (y [N*Ho*Wo, cout], (N, Ho, Wo))"""
        n, h, w = shape
        ho = (h + 2 * self.pad - self.kh) // self.stride + 1
        wo = (w + 2 * self.pad - self.kw) // self.stride + 1
        if self.kh == 1 and self.kw == 1 and self.stride == 1 and self.pad == 0:
            a = x
        else:
            a = torch.empty(n * ho * wo, self.kp, device=x.device, dtype=torch.float32)
            check(_lib.load().clipfs_im2col_nhwc(x.data_ptr(), a.data_ptr(), n, h, w, self.cin, self.kh, self.kw, self.stride,
                                                 self.pad, self.kp, _stream()), "im2col_nhwc")
        y = ops.gemm_nt(a, self.weight, bias=self.bias, residual=residual, act=3 if relu else 0)
        return y, (n, ho, wo)


class MocoResNet50:
    """``model, feat_dim = load_moco(path)``: callable on NORMALISED images [B, 3, H, W] (``tfm_moco``), returns the
    2048-d pooled features (``model.fc`` = Identity, slow_pace.py:1269-1270).  Forward only (the branch is frozen)."""

    feat_dim = 2048

    def __init__(self, state_dict: Dict[str, torch.Tensor], device):
        self.device = torch.device(device)
        sd = state_dict
        self.conv1 = _Conv(sd, "conv1", "bn1", 2, 3, self.device)
        self.blocks: List[dict] = []
        inplanes = 64
        for li, (nb, planes) in enumerate(zip(LAYERS, PLANES)):
            for bi in range(nb):
                stride = 2 if (bi == 0 and li > 0) else 1
                p = f"layer{li + 1}.{bi}"
                blk = {"c1": _Conv(sd, p + ".conv1", p + ".bn1", 1, 0, self.device),
                       "c2": _Conv(sd, p + ".conv2", p + ".bn2", stride, 1, self.device),
                       "c3": _Conv(sd, p + ".conv3", p + ".bn3", 1, 0, self.device), "down": None}
                if bi == 0:
                    blk["down"] = _Conv(sd, p + ".downsample.0", p + ".downsample.1", stride, 0, self.device)
                self.blocks.append(blk)
                inplanes = planes * 4
        self.training = False

    def eval(self):
        return self

    def train(self, mode: bool = True):  # BatchNorm statistics are frozen (module docstring)
        return self

    @torch.no_grad()
    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        x = images.to(self.device, torch.float32).contiguous()
        assert x.dim() == 4 and x.shape[1] == 3, "images: [B, 3, H, W]"
        n, _, h, w = x.shape
        lib = _lib.load()
        xh = torch.empty(n * h * w, 3, device=self.device, dtype=torch.float32)
        check(lib.clipfs_nchw_to_nhwc(x.data_ptr(), xh.data_ptr(), n, 3, h, w, _stream()), "nchw_to_nhwc")
        y, shp = self.conv1(xh, (n, h, w), relu=True)
        n, h, w = shp
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        p = torch.empty(n * ho * wo, 64, device=self.device, dtype=torch.float32)
        check(lib.clipfs_maxpool3x3s2_nhwc(y.data_ptr(), p.data_ptr(), n, h, w, 64, _stream()), "maxpool")
        x, shp = p, (n, ho, wo)
        for blk in self.blocks:
            idn = x
            y, s1 = blk["c1"](x, shp, relu=True)
            y, s2 = blk["c2"](y, s1, relu=True)
            if blk["down"] is not None:
                idn, _ = blk["down"](x, shp, relu=False)
            x, shp = blk["c3"](y, s2, relu=True, residual=idn)  # relu(bn3(conv3) + identity)
        n, h, w = shp
        out = torch.empty(n, x.shape[1], device=self.device, dtype=torch.float32)
        check(lib.clipfs_global_avgpool_nhwc(x.data_ptr(), out.data_ptr(), n, h * w, x.shape[1], _stream()), "avgpool")
        return out

    forward = execute = __call__


def strip_moco_prefix(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """slow_pace.py:1246-1261: ``base_encoder.X`` -> ``X`` (except ``base_encoder.fc*``); also accepts the
    ``module.base_encoder.`` prefix of the original MoCo-v3 checkpoints (moco.py:17-20)."""
    out = {}
    for k, v in state_dict.items():
        for pre in ("module.base_encoder.", "base_encoder."):
            if k.startswith(pre) and not k.startswith(pre + "fc"):
                out[k[len(pre):]] = v
                break
        else:
            out[k] = v
    return out


def synth_resnet50_state_dict(seed: int = 7) -> Dict[str, torch.Tensor]:
    """Random ResNet-50 weights with torchvision key names (kaiming-normal convolutions, perturbed BatchNorm affine and
    running statistics so that the folding is exercised).  The MoCo-v3 checkpoint is not available offline."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cout * k * k)) ** 0.5

    def bn(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.05 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    inplanes = 64
    for li, (nb, planes) in enumerate(zip(LAYERS, PLANES)):
        for bi in range(nb):
            p = f"layer{li + 1}.{bi}"
            conv(p + ".conv1", planes, inplanes, 1)
            bn(p + ".bn1", planes)
            conv(p + ".conv2", planes, planes, 3)
            bn(p + ".bn2", planes)
            conv(p + ".conv3", planes * 4, planes, 1)
            bn(p + ".bn3", planes * 4)
            if bi == 0:
                conv(p + ".downsample.0", planes * 4, inplanes, 1)
                bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    return sd
