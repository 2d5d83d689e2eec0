"""GPU view generation (csrc/views.hip) against PIL itself: the kernel restates Pillow's 8-bit resampling, so the
uint8 pixels must be identical to Image.crop(box).resize(size, BILINEAR / BICUBIC) and the normalised floats
equal to the reference's ImageNormalize formula.  PIL is the library the reference's CPU workers call
(ood.py:946-958 via jittor.transform); it is not part of the oracle."""
import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _pil_views(arr, recs, size):
    from PIL import Image
    from clipfs.views import CLIP_MEAN, CLIP_STD
    img = Image.fromarray(arr)
    mean = np.float32(CLIP_MEAN).reshape(-1, 1, 1)
    std = np.float32(CLIP_STD).reshape(-1, 1, 1)
    outs, raws = [], []
    for top, left, h, w, flip, ow, oh, wx, wy, filt in recs:
        im = img.crop((left, top, left + w, top + h))
        im = im.resize((ow, oh), Image.BICUBIC if filt == 1 else Image.BILINEAR)
        im = im.crop((wx, wy, wx + size, wy + size))
        if flip:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        u8 = np.asarray(im).transpose(2, 0, 1)
        raws.append(u8)
        outs.append((u8 - mean * np.float32(255.)) * (np.float32(1. / 255.) / std))
    return np.stack(outs).astype(np.float32), np.stack(raws)


@gpu
@pytest.mark.parametrize("H,W", [(375, 500), (500, 333), (224, 224), (256, 300)])
def test_views_match_pil_bit_exactly(dev, H, W):
    from clipfs import views
    rng = np.random.RandomState(H + W)
    # smooth + noisy content so interpolation errors would show
    yy, xx = np.mgrid[0:H, 0:W]
    arr = np.stack([(np.sin(xx / 17.0) * 90 + 128), (np.cos(yy / 23.0) * 90 + 128), ((xx + yy) % 256)], -1)
    arr = np.clip(arr + rng.randint(-30, 30, arr.shape), 0, 255).astype(np.uint8)
    recs = views.view_records(W, H, 40, scale=(0.2, 1.0), seed=3)
    assert recs.shape == (41, 10) and recs[0, 9] == views.BICUBIC and (recs[1:, 9] == views.BILINEAR).all()
    got = views.make_views(torch.from_numpy(arr).to(dev), recs).cpu().numpy()
    want, raw = _pil_views(arr, recs, 224)
    # recover the uint8 pixels from the normalised floats and demand exact equality with Pillow
    mean = np.float32(views.CLIP_MEAN).reshape(1, -1, 1, 1)
    std = np.float32(views.CLIP_STD).reshape(1, -1, 1, 1)
    rec_u8 = np.rint(got * std * 255.0 + mean * 255.0).astype(np.int64)
    assert np.array_equal(rec_u8, raw.astype(np.int64)), np.abs(rec_u8 - raw).max()
    assert np.abs(got - want).max() < 1e-5


@gpu
def test_centre_view_is_the_reference_preprocess(dev):
    """view 0 == jclip.clip._transform2 (Resize 256 bicubic, CenterCrop 224, normalise) of the same image."""
    from PIL import Image
    from clipfs import views
    from jclip import clip
    rng = np.random.RandomState(1)
    arr = rng.randint(0, 255, (360, 480, 3), dtype=np.uint8)
    recs = views.view_records(480, 360, 0)
    got = views.make_views(torch.from_numpy(arr).to(dev), recs)[0].cpu()
    want = clip._transform2(224)(Image.fromarray(arr))
    assert (got - want).abs().max() < 1e-5


def test_box_sampler_properties():
    from clipfs import views
    rng = np.random.RandomState(0)
    areas = []
    for _ in range(2000):
        top, left, h, w = views.sample_crop(500, 375, (0.5, 1.0), (3 / 4, 4 / 3), rng)
        assert 0 <= top and top + h <= 375 and 0 <= left and left + w <= 500 and h > 0 and w > 0
        areas.append(h * w / (500 * 375))
    assert 0.45 < min(areas) and max(areas) <= 1.0 and 0.6 < np.mean(areas) < 0.85
    a = views.view_records(500, 375, 8, seed=5)
    b = views.view_records(500, 375, 8, seed=5)
    assert np.array_equal(a, b) and not np.array_equal(a, views.view_records(500, 375, 8, seed=6))
    with pytest.raises(ValueError):
        views.view_records(6000, 6000, 1, scale=(1.0, 1.0), ratio=(1.0, 1.0))
