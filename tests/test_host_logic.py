"""Host-side drop-in API without a GPU: model construction from a state dict (hyper-parameter inference,
jclip/model.py:235-274), apply_lora placement / order, parameter filters, checkpoint schema and errors.
Parameters live on the CPU here; no kernel is launched."""
import os
import types

import numpy as np
import pytest
import torch


def _args(**kw):
    d = dict(encoder="both", position="all", backbone="ViT-B/32", params=["q", "k", "v"], r=4, alpha=1, dropout_rate=0.25)
    d.update(kw)
    return types.SimpleNamespace(**d)


@pytest.fixture(scope="module")
def b32():
    from clipfs import synth
    from jclip.model import build_model
    sd = synth.synth_state_dict(synth.VIT_B32, seed=1)
    sd = dict(sd, input_resolution=torch.tensor(224), context_length=torch.tensor(77), vocab_size=torch.tensor(49408))
    return build_model(sd, device=torch.device("cpu"))


def test_build_model_infers_hyperparameters(b32):
    m = b32
    assert m.visual.input_resolution == 224 and m.visual.patch_size == 32 and m.visual.width == 768
    assert m.visual.transformer.layers == 12 and m.visual.transformer.heads == 12 and m.visual.tokens == 50
    assert m.transformer.width == 512 and m.transformer.heads == 8 and m.transformer.layers == 12
    assert m.context_length == 77 and m.vocab_size == 49408 and m.embed_dim == 512
    assert m.dtype == torch.float32 and not m.training
    assert type(m.transformer.resblocks[0].attn).__name__ == "MultiheadAttention"


def test_apply_lora_order_filters_and_shipped_checkpoint(b32, golden_dir):
    import lora_train_vlp as L
    args = _args()
    layers = L.apply_lora(args, b32)
    assert len(layers) == 24
    assert [l.embed_dim for l in layers] == [512] * 12 + [768] * 12  # text blocks first (lora_train_vlp.py:519-546)
    assert L.apply_lora(args, b32) == []  # already adapted blocks are skipped (class-name check :526)
    params = L.get_lora_parameters(b32)
    assert len(params) == 144 and sum(p.numel() for p in params) == 368640
    names = [n for n, _ in b32.named_parameters() if "lora_" in n]
    assert "transformer.resblocks.0.attn.q_proj.w_lora_A" in names
    assert "visual.transformer.resblocks.11.attn.v_proj.w_lora_B" in names
    L.mark_only_lora_as_trainable(b32)
    assert all(p.requires_grad == ("lora_" in n) for n, p in b32.named_parameters())
    assert set(L.lora_state_dict(b32)) == set(names)
    # bias modes (lora_train_vlp.py:149-160): 'all' re-enables every *bias*, 'lora_only' the biases of the wrapped linears
    L.mark_only_lora_as_trainable(b32, bias="all")
    assert all(p.requires_grad == ("lora_" in n or "bias" in n) for n, p in b32.named_parameters())
    L.mark_only_lora_as_trainable(b32, bias="lora_only")
    on = {n for n, p in b32.named_parameters() if p.requires_grad and "lora_" not in n}
    assert "transformer.resblocks.0.attn.q_proj.bias" in on and "visual.transformer.resblocks.3.attn.v_proj.bias" in on
    assert not any("mlp" in n or "ln_" in n or "out_proj" in n for n in on)
    with pytest.raises(NotImplementedError):
        L.mark_only_lora_as_trainable(b32, bias="some")
    L.mark_only_lora_as_trainable(b32)
    l0 = layers[0]
    assert l0.scaling == 0.5 and l0.q_proj.r == 4  # alpha / sqrt(r)
    assert torch.count_nonzero(l0.q_proj.w_lora_B) == 0  # B = 0 at init (:213)
    assert l0.q_proj.w_lora_A.abs().max() <= 1 / np.sqrt(512) + 1e-7  # kaiming_uniform(a=sqrt 5) bound
    # q/k/v weights are views of the packed in-projection (rows [0:d],[d:2d],[2d:3d], :395-409)
    assert l0.k_proj.weight.data_ptr() == l0.qkv_weight[512:].data_ptr()
    L.load_lora(args, layers, os.path.join(golden_dir, "lora_weights.pkl"))
    from clipfs import safe_pkl
    ck = safe_pkl.load(os.path.join(golden_dir, "lora_weights.pkl"))
    assert np.array_equal(layers[13].v_proj.w_lora_B.detach().numpy(), ck["weights"]["layer_13"]["v_proj"]["w_lora_B"])
    assert np.array_equal(layers[13].lora_A_qkv[4:8].numpy(), ck["weights"]["layer_13"]["k_proj"]["w_lora_A"])
    for field, val in (("r", 8), ("alpha", 2), ("encoder", "text"), ("params", ["q"]), ("position", "up")):
        with pytest.raises(ValueError, match="mismatch"):
            L.load_lora(_args(**{field: val}), layers, os.path.join(golden_dir, "lora_weights.pkl"))
    with pytest.raises(FileNotFoundError):
        L.load_lora(args, layers, "/nonexistent/lora.pkl")


def test_save_lora_schema_roundtrip(tmp_path):
    import lora_train_vlp as L
    from clipfs import safe_pkl, synth
    from jclip.model import build_model
    m = build_model(synth.synth_state_dict(synth.TINY, seed=2), device=torch.device("cpu"))
    args = _args(backbone="tiny", position="bottom", params=["q", "v", "o"], r=2, encoder="vision")
    L.INDEX_POSITIONS_VISION["tiny"] = {"bottom": [0, 1]}
    try:
        layers = L.apply_lora(args, m)
    finally:
        del L.INDEX_POSITIONS_VISION["tiny"]
    assert len(layers) == 2 and layers[0].lora_mask == 1 | 4 | 8
    with torch.no_grad():
        layers[1].proj.w_lora_B.normal_()
    path = str(tmp_path / "lora_weights1" / "lora_weights.pkl")
    L.save_lora(args, 3, layers, save_path=path)
    ck = safe_pkl.load(path)
    assert ck["metadata"] == {"r": 2, "alpha": 1, "encoder": "vision", "params": ["q", "v", "o"], "position": "bottom"}
    assert sorted(ck["weights"]["layer_1"]) == ["proj", "q_proj", "v_proj"]
    assert np.array_equal(ck["weights"]["layer_1"]["proj"]["w_lora_B"], layers[1].proj.w_lora_B.detach().numpy())
    assert type(layers[0].k_proj).__name__ == "_FrozenLinear"


def test_engine_refuses_cpu_tensors(b32):
    with pytest.raises((AssertionError, RuntimeError)):
        b32.encode_image(torch.zeros(1, 3, 224, 224))


def test_shard_bounds():
    from clipfs.dist import shard_bounds
    for n, w in ((403, 8), (256, 8), (7, 3), (3, 8)):
        parts = [shard_bounds(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1


def test_clip_load_checkpoint_formats(tmp_path):
    """clip.load / clip1.load_vlp from the three accepted local formats: a Jittor-style pickle of numpy arrays
    (read without executing anything), .npz, and a torch zip checkpoint; 5-tuple return like the reference."""
    import pickle
    from clipfs import synth
    from jclip import clip, clip1
    sd = synth.synth_state_dict(synth.TINY, seed=3)
    np_sd = {k: v.numpy() for k, v in sd.items()}
    np_sd.update(input_resolution=np.array(64), context_length=np.array(16), vocab_size=np.array(512))
    p_pkl, p_npz, p_pt = tmp_path / "m.pkl", tmp_path / "m.npz", tmp_path / "m.pt"
    with open(p_pkl, "wb") as f:
        pickle.dump(np_sd, f, protocol=4)
    np.savez(p_npz, **{k: v for k, v in np_sd.items()})
    torch.save(sd, p_pt)
    cpu = torch.device("cpu")
    for path in (p_pkl, p_npz, p_pt):
        out = clip.load(str(path), device=cpu)
        assert len(out) == 5
        model = out[0]
        assert model.visual.width == 128 and model.visual.tokens == 5 and model.visual.VPT is None
        assert torch.equal(model.text_projection.data, sd["text_projection"])
        assert not hasattr(model, "input_resolution") or True
    mv = clip1.load_vlp(str(p_pkl), device=cpu)[0]
    assert mv.visual.VPT.shape == (4, 128) and mv.visual.tokens == 9  # 4 VPT tokens after the patches
    assert "visual.VPT" in dict(mv.named_parameters())
    with pytest.raises(NotImplementedError):
        clip.load(str(p_pkl), mode="res", device=cpu)


def test_transforms_shapes():
    from PIL import Image
    from clipfs import synth
    from jclip import clip
    from jclip.model import build_model
    rng = np.random.RandomState(0)
    img = Image.fromarray(rng.randint(0, 255, (300, 400, 3), dtype=np.uint8))
    t1, t2, t3, t4 = clip._transform1(224), clip._transform2(224), clip.tfm_train_base(224), clip.tfm_train_base1(224)
    a, b = t1(img), t2(img)
    assert a.shape == (3, 224, 224) and a.dtype == torch.float32 and 0 <= a.min() and a.max() <= 1
    mean = torch.tensor(clip.CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(clip.CLIP_STD).view(3, 1, 1)
    assert torch.allclose(b, (a - mean) / std, atol=1e-6)
    assert t3(img).shape == t4(img).shape == (3, 224, 224)


def test_stage2_module_checkpoints_roundtrip(tmp_path):
    """slow_pace.py:1709-1713 / test.py:1818-1821: ``channel_lp.save``, ``prompt_learner.save``, ``clip_model.save`` write
    {dotted name: array} pickles; ``.load`` copies them back in place (views / flat buffers stay bound) and the files are
    plain dicts readable without executing anything."""
    import types
    import lora_train_vlp as L
    import slow_pace as S
    from clipfs import safe_pkl, synth
    from jclip.model import build_model
    cpu = torch.device("cpu")
    cfg = synth.TINY

    def make(seed):
        model = build_model(synth.synth_state_dict(cfg, seed=seed, perturb=True), device=cpu)
        args = types.SimpleNamespace(encoder="both", position="all", backbone="tiny", params=["q", "v", "o"], r=2, alpha=1,
                                     dropout_rate=0.0)
        saved = L.INDEX_POSITIONS_TEXT["all"]
        L.INDEX_POSITIONS_TEXT["all"] = list(range(cfg.transformer_layers))
        L.INDEX_POSITIONS_VISION["tiny"] = {"all": list(range(cfg.vision_layers))}
        try:
            layers = L.apply_lora(args, model)
        finally:
            L.INDEX_POSITIONS_TEXT["all"] = saved
            del L.INDEX_POSITIONS_VISION["tiny"]
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for layer in layers:
                for p_, _ in layer.trainable_pairs():
                    p_.copy_(torch.randn(p_.shape, generator=g) * 0.05)
        return model, layers

    model, layers = make(3)
    path = str(tmp_path / "test_pkl" / "clip_model.pkl")
    model.save(path)
    raw = safe_pkl.load(path)
    assert isinstance(raw, dict) and all(isinstance(v, np.ndarray) for v in raw.values())
    assert "transformer.resblocks.0.attn.q_proj.w_lora_A" in raw and "visual.conv1.weight" in raw
    assert "transformer.resblocks.0.attn.k_proj.weight" in raw and "transformer.resblocks.0.attn.proj.w_lora_B" in raw
    other, other_layers = make(4)
    before = other_layers[0].lora_A_qkv.clone()
    other.load(path)
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), other.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2), n1
    assert not torch.equal(other_layers[0].lora_A_qkv, before)  # the stacked tensor behind the views was written
    assert torch.equal(other_layers[0].lora_A_qkv, layers[0].lora_A_qkv)
    with pytest.raises(FileNotFoundError):
        other.load(str(tmp_path / "nope.pkl"))

    head = S.Channel_LP(in_dim=cfg.embed_dim, n_classes=7, device=cpu)
    with torch.no_grad():
        head.scale1.add_(0.25)
        head.bias1.sub_(0.5)
    hp = str(tmp_path / "test_pkl" / "channel.pkl")
    head.save(hp)
    assert sorted(safe_pkl.load(hp)) == ["bias1", "fc.bias", "fc.weight", "scale1"]
    head2 = S.Channel_LP(in_dim=cfg.embed_dim, n_classes=7, device=cpu)
    head2.load(hp)
    assert all(torch.equal(a, b) for a, b in zip(head.state_dict().values(), head2.state_dict().values()))
    bad = S.Channel_LP(in_dim=cfg.embed_dim, n_classes=9, device=cpu)
    with pytest.raises(ValueError, match="shape mismatch"):
        bad.load(hp)
