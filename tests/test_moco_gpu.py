"""MoCo-v3 ResNet-50 auxiliary branch (slow_pace.py:1208-1219,1237-1274,1151-1168,1542-1552,
1677-1680) on the HIP engine against the oracle's torch restatement, synthetic weights (the pretrained r-50-1000ep.pkl
is not available offline)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _err(a, b):
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()


@pytest.fixture(scope="module")
def r50(dev):
    import slow_pace as SP
    from clipfs import resnet
    sd = resnet.synth_resnet50_state_dict(seed=7)
    # the checkpoint layout the reference reads: {'state_dict': {'base_encoder.<name>': array}} (+ a head it drops)
    ck = {"state_dict": {"base_encoder." + k: v.numpy() for k, v in sd.items()}}
    ck["state_dict"]["base_encoder.fc.weight"] = np.zeros((4, 4), np.float32)
    model, dim = SP.load_moco(ck, device=dev)
    assert dim == 2048
    return sd, model


def test_data_movement_kernels(dev):
    import ctypes as C
    import torch.nn.functional as F
    from clipfs import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 8, 13, 11, generator=g).to(dev)  # NCHW
    xh = torch.empty(3 * 13 * 11, 8, device=dev)
    _lib.check(lib.clipfs_nchw_to_nhwc(x.data_ptr(), xh.data_ptr(), 3, 8, 13, 11, st))
    assert torch.equal(xh.reshape(3, 13, 11, 8), x.permute(0, 2, 3, 1))
    for kh, stride, pad, C_ in ((3, 1, 1, 8), (3, 2, 1, 8), (1, 2, 0, 8), (7, 2, 3, 3)):
        xx = torch.randn(2, C_, 13, 11, generator=g).to(dev)
        nh = xx.permute(0, 2, 3, 1).contiguous()
        ho, wo = (13 + 2 * pad - kh) // stride + 1, (11 + 2 * pad - kh) // stride + 1
        k = kh * kh * C_
        kp = (k + 3) // 4 * 4
        col = torch.full((2 * ho * wo, kp), 7.0, device=dev)
        _lib.check(lib.clipfs_im2col_nhwc(nh.data_ptr(), col.data_ptr(), 2, 13, 11, C_, kh, kh, stride, pad, kp, st))
        want = F.unfold(xx, kh, padding=pad, stride=stride)              # [2, C*kh*kw, L] in (c, ky, kx) order
        want = want.reshape(2, C_, kh * kh, ho * wo).permute(0, 3, 2, 1).reshape(2 * ho * wo, k)  # -> (ky, kx, c)
        assert torch.equal(col[:, :k], want) and (col[:, k:] == 0).all()
    y = torch.randn(2, 16, 9, 7, generator=g).to(dev)
    yh = y.permute(0, 2, 3, 1).contiguous()
    mp = torch.empty(2 * 5 * 4, 16, device=dev)
    _lib.check(lib.clipfs_maxpool3x3s2_nhwc(yh.data_ptr(), mp.data_ptr(), 2, 9, 7, 16, st))
    assert torch.equal(mp.reshape(2, 5, 4, 16), F.max_pool2d(y, 3, 2, 1).permute(0, 2, 3, 1))
    ap = torch.empty(2, 16, device=dev)
    _lib.check(lib.clipfs_global_avgpool_nhwc(yh.data_ptr(), ap.data_ptr(), 2, 63, 16, st))
    assert _err(ap, y.mean(dim=(2, 3))) < 1e-6
    assert lib.clipfs_im2col_nhwc(nh.data_ptr(), col.data_ptr(), 2, 13, 11, 3, 7, 7, 2, 3, 146, st) == 1  # Kp too small


def test_relu_epilogue(dev):
    from clipfs import ops
    g = torch.Generator().manual_seed(2)
    a, b = torch.randn(200, 96, generator=g).to(dev), torch.randn(72, 96, generator=g).to(dev)
    bias, res = torch.randn(72, generator=g).to(dev), torch.randn(200, 72, generator=g).to(dev)
    got = ops.gemm_nt(a, b, bias=bias, residual=res, act=3)
    want = torch.relu(a.double() @ b.double().t() + bias.double() + res.double())
    assert _err(got, want) < 1e-4 and (got >= 0).all() and (got == 0).any()


@pytest.mark.parametrize("res", [64, 224])
def test_resnet50_features_vs_oracle(dev, r50, res):
    from oracle import clip_oracle as O
    sd, model = r50
    B = 3 if res == 64 else 2
    x = torch.randn(B, 3, res, res, generator=torch.Generator().manual_seed(3))
    got = model(x.to(dev))
    want = O.resnet50_forward({k: v.double() for k, v in sd.items()}, x.double())
    assert got.shape == (B, 2048)
    scale = want.abs().max().item()
    assert _err(got, want) < 2e-4 * scale, (_err(got, want), scale)


def test_moco_adapter_init_loss_and_checkpoint(dev, r50, tmp_path):
    import slow_pace as SP
    from oracle import clip_oracle as O
    sd, model = r50
    g = torch.Generator().manual_seed(5)
    n, Cn = 12, 7
    images = torch.rand(n, 3, 64, 64, generator=g)
    labels = torch.tensor([0, 1, 2, 3, 4, 5, 6, 0, 1, 2, 3, 3])
    loader = [(images[:5], labels[:5], None), (images[5:], labels[5:], None)]
    feats, lab = SP.pre_load_features_moco(model, loader)
    sd64 = {k: v.double() for k, v in sd.items()}
    mean = torch.tensor(SP.MOCO_MEAN_STD[0], dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor(SP.MOCO_MEAN_STD[1], dtype=torch.float64).view(1, 3, 1, 1)
    wf = O.resnet50_forward(sd64, (images.double() - mean) / std)
    wf = wf / wf.norm(dim=-1, keepdim=True)
    assert torch.equal(lab.cpu(), labels) and _err(feats, wf) < 2e-5
    ad = SP.Moco_Adapter(2048, Cn, device=dev)
    SP.moco_adapter_init(ad, feats, lab)
    w0 = torch.zeros(Cn, 2048, dtype=torch.float64)
    for i in range(n):
        w0[int(labels[i])] += wf[i]                                      # slow_pace.py:1548-1550
    assert _err(ad.fc.weight, w0) < 5e-5
    # loss_aux and its gradients (slow_pace.py:1677-1680)
    f = model(SP.tfm_moco(images[:6].to(dev)))
    tgt = labels[:6].to(dev)
    out = SP.logit_normalize(ad(f))
    from clipfs import engine as E
    loss = E.cross_entropy_loss(out, tgt)
    loss.backward()
    ow = ad.fc.weight.detach().double().cpu().requires_grad_()
    ob = ad.fc.bias.detach().double().cpu().requires_grad_()
    wl = O.moco_aux_loss(f.double().cpu(), ow, ob, labels[:6])
    wl.backward()
    assert abs(loss.item() - wl.item()) < 1e-4
    assert _err(ad.fc.weight.grad, ow.grad) < 1e-4 * max(ow.grad.abs().max().item(), 1e-3)
    assert _err(ad.fc.bias.grad, ob.grad) < 1e-4 * max(ob.grad.abs().max().item(), 1e-3)
    path = str(tmp_path / "test_pkl" / "moco_adapter.pkl")
    ad.save(path)
    ad2 = SP.Moco_Adapter(2048, Cn, device=dev)
    ad2.load(path)
    assert torch.equal(ad2.fc.weight, ad.fc.weight) and torch.equal(ad2.fc.bias, ad.fc.bias)
    with pytest.raises(FileNotFoundError):
        SP.load_moco(str(tmp_path / "r-50-1000ep.pkl"))
