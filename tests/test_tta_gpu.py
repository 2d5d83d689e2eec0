"""Stage-2 / TTA drop-in modules on the GPU: VLPromptLearner + TextEncoder (slow_pace.py:110-205,828-848),
Channel_LP / logit_normalize, solve_mta (both return conventions), ood.split_ood and tta.fuse_top5 against the
oracle on seeded inputs."""
import types

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
def b32(dev):
    """ViT-B/32-shaped model with the REAL vocabulary (the prompt learner tokenizes real class names)."""
    from clipfs import synth
    from jclip.model import build_model
    import dataclasses
    cfg = dataclasses.replace(synth.VIT_B32, vision_layers=2, transformer_layers=2)
    sd = synth.synth_state_dict(cfg, seed=7, perturb=True)
    return cfg, sd, build_model(sd, device=dev)


def test_prompt_learner_and_text_encoder(dev, b32, golden_dir):
    import os
    import slow_pace as SP
    from jclip import clip
    from oracle import clip_oracle as O
    cfg, sd, model = b32
    names = [ln.split()[0] for ln in open(os.path.join(golden_dir, "classes.txt"))][:11]
    classnames = [n.split("_", 1)[1] if "_" in n else n for n in names]  # dataset prefix stripped (slow_pace.py:1845-1859)
    pl = SP.VLPromptLearner(classnames, model)
    assert pl.ctx.shape == (4, 512) and pl.tokenized_prompts.shape == (11, 77)
    want_ctx = sd["token_embedding.weight"][clip.tokenize("a photo of a")[0, 1:5]]
    assert _err(pl.ctx, want_ctx) == 0  # initialised from the embeddings of "a photo of a" (:124-131)
    assert pl.tokenized_prompts[0, 1:5].tolist() == [320, 1125, 539, 320]
    enc = SP.TextEncoder(model)
    feats = enc(pl(), pl.tokenized_prompts)
    sd64 = {k: v.double() for k, v in sd.items()}
    octx = pl.ctx.detach().double().cpu().requires_grad_()
    prompts = O.build_prompts(octx, sd64["token_embedding.weight"], pl.tokenized_prompts.cpu())
    want = O.encode_text(sd64, pl.tokenized_prompts.cpu(), embeds=prompts)
    assert _err(feats, want) < 2e-5
    # the materialised [C,77,d] prompt tensor of the reference equals the symbolic PromptBatch
    assert _err(pl().materialize(model), prompts) < 1e-6
    # gradient of a scalar of the features with respect to the shared ctx tokens
    w = torch.randn(11, 512, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    (want * w).sum().backward()
    (feats * w.float().to(dev)).sum().backward()
    assert _err(pl.ctx.grad, octx.grad) < 1e-4 * octx.grad.abs().max().item()
    with pytest.raises(TypeError):
        enc(prompts.float().to(dev), pl.tokenized_prompts)


def test_channel_lp_and_logit_normalize(dev):
    import slow_pace as SP
    from oracle import clip_oracle as O
    g = torch.Generator().manual_seed(3)
    head = SP.Channel_LP(512, 403, device=dev)
    with torch.no_grad():
        head.scale1.copy_(1 + 0.1 * torch.randn(512, generator=g))
        head.bias1.copy_(0.1 * torch.randn(512, generator=g))
        head.fc.weight.copy_(torch.randn(403, 512, generator=g) / 512 ** 0.5)  # zero-shot text features go here (:1537-1540)
    f = torch.randn(37, 512, generator=g)
    z = head(f.to(dev))
    want = O.channel_lp(f.double(), head.scale1.detach().double().cpu(), head.bias1.detach().double().cpu(),
                        head.fc.weight.detach().double().cpu(), head.fc.bias.detach().double().cpu())
    assert _err(z, want) < 1e-4
    assert _err(SP.logit_normalize(z), O.logit_normalize(want)) < 1e-4


def test_tta_pipeline_views_to_top5(dev, b32):
    """cfg-4 in miniature: n_img x (1 + N) views -> image tower -> normalise -> MTA -> OOD split / top-5,
    against the oracle run per image (reference loop: ood.py:867-883, test.py:1705-1742)."""
    import lora_train_vlp as L
    import ood
    import slow_pace as SP
    import tta
    from clipfs import ops, synth
    from oracle import clip_oracle as O
    cfg, sd, model = b32
    n_img, V, Cn = 2, 9, 21
    g = torch.Generator().manual_seed(5)
    base = torch.randn(n_img, 1, 3, 224, 224, generator=g)
    views = (base + 0.3 * torch.randn(n_img, V, 3, 224, 224, generator=g)).contiguous()
    text = O.l2_normalize(torch.randn(Cn, 512, generator=g, dtype=torch.float64)).float()
    sd64 = {k: v.double() for k, v in sd.items()}
    logits, mode = ood.mta_scores(model, views.to(dev), text.to(dev), want_mode=True)
    is_base, pred = ood.split_ood(model, views.to(dev), text.to(dev))
    for i in range(n_img):
        f = O.l2_normalize(O.encode_image(sd64, views[i].double())).float()
        wl = O.solve_mta(f, text.t())
        wm = O.solve_mta(f, text.t(), return_mode=True)
        assert _err(logits[i:i + 1], wl) < 5e-3
        assert _err(mode[i:i + 1], wm) < 5e-5
        assert int(pred[i]) == int(wl.argmax())
        assert bool(is_base[i]) == bool(O.ood_is_base(wl)[0])
        # the two reference entry points on one image
        fg = ops.l2norm_fwd(model.encode_image(views[i].to(dev)).contiguous())
        assert _err(L.solve_mta(fg, text.t().to(dev)), wl) < 5e-3
        assert _err(SP.solve_mta(fg, text.t().to(dev)), wm) < 5e-5
    # fusion (test.py:1729-1742)
    cos, cos1, cos3 = [torch.randn(n_img, Cn, generator=g) for _ in range(3)]
    head = torch.randn(n_img, Cn, generator=g)
    got = tta.fuse_top5(cos.to(dev), cos1.to(dev), cos3.to(dev), head.to(dev))
    want = O.fuse_top5(cos, cos1, cos3, head)
    assert torch.equal(got["top5"].cpu().long(), want["top5"])
    assert _err(got["cos5"], want["cos5"]) < 1e-6 and _err(got["cos4"], want["cos4"]) < 1e-6
    out = tta.evaluate_views(model, model, views.to(dev), text.to(dev), text.to(dev), text.to(dev))
    assert out["top5"].shape == (n_img, 5)


def test_score_stream_is_the_per_group_scoring(dev, b32):
    """ood.score_stream (views | tower | MTA on three HIP streams, one group ahead) returns exactly what the one-stream
    loop returns group by group: same kernels, same seeds, only the order in which the GPU sees them differs -- checked
    over 3 groups (ragged last one) and again on a second call (the side streams are new each time)."""
    import numpy as np
    import ood
    import tta
    from oracle import clip_oracle as O
    cfg, sd, model = b32
    rng = np.random.RandomState(11)
    srcs = [torch.from_numpy(rng.randint(0, 256, (60 + 7 * i, 90 - 5 * i, 3), dtype=np.uint8)) for i in range(5)]
    g = torch.Generator().manual_seed(2)
    text = O.l2_normalize(torch.randn(33, 512, generator=g, dtype=torch.float64)).float().to(dev)
    want5, wantb, wantl = [], [], []
    for lo in range(0, 5, 2):
        views = torch.stack([tta.make_tta_views(srcs[i], 6, seed=40 + i, device=dev) for i in range(lo, min(lo + 2, 5))])
        t5, ib, lg = ood.score_views(model, views, text)
        want5.append(t5), wantb.append(ib), wantl.append(lg)
    want5, wantb, wantl = torch.cat(want5), torch.cat(wantb), torch.cat(wantl)
    for _ in range(2):
        t5, ib, lg = ood.score_stream(model, srcs, text, n_crops=6, images_per_pass=2, seed=40)
        torch.cuda.synchronize()
        assert torch.equal(t5, want5) and torch.equal(ib, wantb) and torch.equal(lg, wantl)
    with pytest.raises(ValueError):
        ood.score_stream(model, [], text)


def test_clip_classifier_and_cls_acc(dev, b32):
    import lora_train_vlp as L
    import ood
    from jclip import clip
    from oracle import clip_oracle as O
    cfg, sd, model = b32
    templates = {0: ["a photo of a cat.", "a picture of a cat."], 1: ["a photo of a dog.", "a picture of a dog."],
                 2: ["a photo of a bear.", "a picture of a bear."]}
    w = L.clip_classifier(templates, model)
    assert w.shape == (1, 3, 512)  # callers do .squeeze(0).t() (lora_train_vlp.py:925-926)
    sd64 = {k: v.double() for k, v in sd.items()}
    texts = [t for c in templates for t in templates[c]]
    emb = O.encode_text(sd64, clip.tokenize(texts))
    want = O.class_text_features(emb, [0, 0, 1, 1, 2, 2], 3)  # [d, C]
    assert _err(w.squeeze(0).t(), want) < 2e-5
    out = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1], [0.2, 0.3, 0.5], [0.6, 0.3, 0.1]])
    tgt = torch.tensor([1, 0, 1, 2])
    assert L.cls_acc(out.to(dev), tgt) == O.cls_acc(out, tgt) == 50.0
    assert L.cls_acc(out.to(dev), tgt, topk=2) == O.cls_acc(out, tgt, topk=2) == 75.0
    big = torch.full((3, 403), -1.0)
    big[0, 372] = 1
    big[1, 373] = 1
    big[2, 5] = 1
    t2 = torch.tensor([3, 400, 390])
    assert ood.cls_acc(big.to(dev), t2) == O.cls_acc_ood(big, t2)


def test_head_training_step(dev):
    """Stage-2 head objective (slow_pace.py:1666-1675): CE(logit_normalize(Channel_LP(concat(img feats, text feats))),
    concat(target, arange(C))) -- loss and gradients of scale1 / bias1 / fc.weight / fc.bias vs fp64 autograd."""
    import slow_pace as SP
    from clipfs import engine as E
    from oracle import clip_oracle as O
    g = torch.Generator().manual_seed(11)
    Cn, d, B = 403, 512, 32
    head = SP.Channel_LP(d, Cn, device=dev)
    txt = O.l2_normalize(torch.randn(Cn, d, generator=g, dtype=torch.float64))
    with torch.no_grad():
        head.fc.weight.copy_(txt.float())                      # zero-shot text features (:1537-1540)
        head.scale1.copy_(1 + 0.05 * torch.randn(d, generator=g))
        head.bias1.copy_(0.05 * torch.randn(d, generator=g))
    feats = torch.cat([torch.randn(B, d, generator=g) * 0.4, txt.float()], dim=0)   # raw image feats + text feats
    target = torch.cat([torch.randint(0, 374, (B,), generator=g), torch.arange(Cn)])
    out = SP.logit_normalize(head(feats.to(dev)))
    loss = E.cross_entropy_loss(out, target.to(dev))
    loss.backward()
    p64 = [t.detach().double().cpu().requires_grad_() for t in (head.scale1, head.bias1, head.fc.weight, head.fc.bias)]
    want = O.jt_cross_entropy(O.logit_normalize(O.channel_lp(feats.double(), *p64)), target)
    want.backward()
    assert abs(loss.item() - want.item()) < 1e-5
    for mine, ref in zip((head.scale1, head.bias1, head.fc.weight, head.fc.bias), p64):
        err = (mine.grad.double().cpu() - ref.grad).abs().max().item()
        assert err < 1e-4 * max(ref.grad.abs().max().item(), 1e-3), err


def test_evaluate_lora_three_accuracies(dev, b32):
    """lora_train_vlp.py:813-846: MTA / centre-view / view-ensemble top-1 accuracy from ONE encode pass, against the
    oracle run image by image exactly as the reference loop does (loader batch size 1, views = centre + crops)."""
    import lora_train_vlp as L
    from oracle import clip_oracle as O
    cfg, sd, model = b32
    n_img, N, Cn = 5, 6, 13
    g = torch.Generator().manual_seed(11)
    base = torch.randn(n_img, 1, 3, 224, 224, generator=g)
    centre = (base + 0.2 * torch.randn(n_img, 1, 3, 224, 224, generator=g)).contiguous()
    crops = (base + 0.6 * torch.randn(n_img, N, 3, 224, 224, generator=g)).contiguous()
    text = O.l2_normalize(torch.randn(Cn, 512, generator=g, dtype=torch.float64)).float()  # [C, d]
    sd64 = {k: v.double() for k, v in sd.items()}
    want = [0, 0, 0]
    target = []
    for i in range(n_img):
        views = torch.cat([centre[i], crops[i]], 0)
        f = O.l2_normalize(O.encode_image(sd64, views.double())).float()
        mta = O.solve_mta(f, text.t())
        basel = f[0:1] @ text.t()
        ens = (f @ text.t()).mean(dim=0, keepdim=True)
        tgt = int(mta.argmax()) if i % 2 == 0 else int((mta.argmax() + 1) % Cn)  # a mix of hits and misses
        target.append(tgt)
        for k, x in enumerate((mta, basel, ens)):
            want[k] += int(int(x.argmax()) == tgt)
    # the reference's loader: (image [1,1,3,R,R], images [1,1,N,3,R,R], target, impath), one image per batch
    loader = [(centre[i].unsqueeze(0), crops[i].unsqueeze(0).unsqueeze(0), torch.tensor([target[i]]), "x") for i in range(n_img)]
    got = L.evaluate_lora(None, model, loader, textual_features=text.t().to(dev))
    assert got == tuple(100.0 * w / n_img for w in want), (got, want)
    # batched loader (several images per batch) gives the same numbers
    loader2 = [(centre[:3], crops[:3], torch.tensor(target[:3]), "x"), (centre[3:], crops[3:], torch.tensor(target[3:]), "x")]
    assert L.evaluate_lora(None, model, loader2, textual_features=text.t().to(dev)) == got
    with pytest.raises(ValueError):
        L.evaluate_lora(None, model, loader)


def test_prompt_learner_checkpoint_roundtrip(dev, b32, golden_dir, tmp_path):
    """prompt_learner.save / .load (slow_pace.py:1712, test.py:1821)."""
    import os
    import slow_pace as SP
    from clipfs import safe_pkl
    cfg, sd, model = b32
    names = [ln.split()[0] for ln in open(os.path.join(golden_dir, "classes.txt"))][:5]
    classnames = [n.split("_", 1)[1] if "_" in n else n for n in names]
    pl = SP.VLPromptLearner(classnames, model)
    with torch.no_grad():
        pl.ctx.add_(0.01 * torch.randn(4, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(2)))
    path = str(tmp_path / "test_pkl" / "PromptLearner.pkl")
    pl.save(path)
    raw = safe_pkl.load(path)
    assert set(raw) == {"ctx", "token_prefix", "token_suffix", "tokenized_prompts"}
    assert raw["token_prefix"].shape == (5, 1, 512) and raw["token_suffix"].shape == (5, 72, 512)
    pl2 = SP.VLPromptLearner(classnames, model)
    assert _err(pl2.ctx, pl.ctx) > 0
    pl2.load(path)
    assert _err(pl2.ctx, pl.ctx) == 0


def test_adapted_attention_block_direct_call(dev, b32):
    """PlainMultiheadAttentionLoRA.execute on [L, N, d] (lora_train_vlp.py:431-513) vs the oracle's mha_forward with the
    same adapters, causal and not, q/k/v/o adapters, eval mode (no dropout); and train mode drops (output changes)."""
    import lora_train_vlp as L
    from oracle import clip_oracle as O
    cfg, sd, model = b32
    blk = model.transformer.resblocks[1]
    plain = blk.attn
    mha = L.PlainMultiheadAttentionLoRA(plain, enable_lora=["q", "k", "v", "o"], r=4, lora_alpha=1, dropout_rate=0.25, seed=3)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        mha.lora_B_qkv.copy_(0.05 * torch.randn(mha.lora_B_qkv.shape, generator=g))
        mha.lora_B_o.copy_(0.05 * torch.randn(mha.lora_B_o.shape, generator=g))
    d = 512
    Lq, N = 77, 3
    x = torch.randn(Lq, N, d, generator=g)
    p64 = {k: v.double() for k, v in O._block_params(sd, "transformer", 1).items()}
    names = {"q": "q_proj", "k": "k_proj", "v": "v_proj", "o": "proj"}
    lora = {names[k]: {"w_lora_A": getattr(mha, names[k]).w_lora_A.detach().double().cpu(),
                       "w_lora_B": getattr(mha, names[k]).w_lora_B.detach().double().cpu()} for k in "qkvo"}
    mha.eval()
    for causal in (False, True):
        mask = O.build_causal_mask(Lq, torch.float64) if causal else None
        want = O.mha_forward(x.double(), p64, 8, mask, lora, O.lora_scaling(1, 4))
        got, w = mha(x.to(dev), x.to(dev), x.to(dev), need_weights=False, attn_mask=mask)
        assert w is None and _err(got, want) < 5e-5
    mha.train()
    dropped, _ = mha(x.to(dev), attn_mask=None)
    assert _err(dropped, want) > 1e-4
    with pytest.raises(NotImplementedError):
        mha(x.to(dev), need_weights=True)
