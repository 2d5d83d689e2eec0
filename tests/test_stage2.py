"""Stage-2 objective (slow_pace.py:1590-1697 without the MoCo branch): oracle restatement against torch.nn.functional
(CPU), the HIP loss kernels and the assembled ``stage2_loss`` against the oracle (GPU).  Jittor itself is not
installable here: the Jittor semantics (nn.l1_loss = mean abs; kl_div as written at slow_pace.py:1170-1177;
CosineAnnealingLR closed form) are restated from the reference's call sites -- parity unpinned against Jittor."""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "jittor-clip-fewshot_amd"))

from oracle import clip_oracle as O  # noqa: E402


def _inputs(B=6, C=9, d=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    unit = lambda t: t / t.norm(dim=-1, keepdim=True)
    return dict(img=r(B, d), txt=r(C, d), target=torch.randint(0, C, (B,), generator=g), zs_img=unit(r(B, d)),
                zs_txt=unit(r(C, d)), lp_img=r(B, d), lp_txt=unit(r(C, d)),
                lp=(1 + 0.1 * r(d), 0.1 * r(d), unit(r(C, d)), 0.01 * r(C)))


def test_oracle_losses_match_torch_functional():
    x = _inputs()
    a, b = x["img"] @ x["txt"].t(), x["zs_img"] @ x["zs_txt"].t() * 100
    assert abs(O.jt_l1_loss(x["img"], x["zs_img"]) - F.l1_loss(x["img"], x["zs_img"])) < 1e-15
    la, lb = F.log_softmax(a, 1), F.log_softmax(b, 1)
    assert abs(O.kl_div(la, lb) - F.kl_div(la, lb, reduction="sum", log_target=True)) < 1e-12
    assert abs(O.scl_logits_loss(a, b) - F.kl_div(la, lb, reduction="sum", log_target=True) / a.numel()) < 1e-14
    assert O.kl_div(la, la).abs() < 1e-15 and O.scl_logits_loss(a, b) > 0


def test_oracle_stage2_loss_is_the_sum_of_its_terms():
    x = _inputs()
    loss, terms, cos = O.stage2_loss(x["img"], x["txt"], x["target"], x["zs_img"], x["zs_txt"], x["lp"], x["lp_img"], x["lp_txt"])
    assert abs(loss - sum(terms.values())) < 1e-12
    assert cos.shape == (6, 9) and cos.abs().max() <= 100 + 1e-9
    # head targets are concat(target, arange(C)) (:1667-1668): the C text rows are classified as themselves
    feats = torch.cat((x["lp_img"], x["lp_txt"]))
    out = O.logit_normalize(O.channel_lp(feats, *x["lp"]))
    tgt = torch.cat((x["target"], torch.arange(9)))
    assert abs(terms["lp_ce"] - F.cross_entropy(out, tgt)) < 1e-12


def test_cosine_annealing_closed_form_matches_torch_scheduler():
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=2e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 20, eta_min=1e-6)  # slow_pace.py:1591
    for t in range(1, 21):
        opt.step()
        sch.step()
        assert abs(sch.get_last_lr()[0] - O.cosine_annealing_lr(2e-4, t, 20)) < 1e-15
    assert abs(O.cosine_annealing_lr(2e-4, 20, 20) - 1e-6) < 1e-18 and O.cosine_annealing_lr(2e-4, 0, 20) == 2e-4


def test_mirror_scheduler_matches_oracle():
    import slow_pace as S
    sch = S.CosineAnnealingLR(2e-4, 20)
    assert sch.get_lr() == 2e-4
    for t in range(1, 45):  # stepped per iteration in the reference (:1697): runs past T_max, periodic
        assert abs(sch.step() - O.cosine_annealing_lr(2e-4, t, 20)) < 1e-18


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(403, 512), (7, 33), (256, 403)])
def test_l1_and_kl_kernels(shape):
    from clipfs import ops
    g = torch.Generator().manual_seed(shape[0])
    a = torch.randn(*shape, generator=g)
    b = torch.randn(*shape, generator=g)
    b[0, :3] = a[0, :3]  # exact ties: sign(0) = 0
    ad = a.double().requires_grad_()
    ref = O.jt_l1_loss(ad, b.double())
    ref.backward()
    loss, da = ops.l1_loss(a.cuda(), b.cuda(), want_grad=True, grad_scale=2.0)
    assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
    assert (da.cpu().double() - 2.0 * ad.grad).abs().max() < 1e-9
    x = (a * 10).double().requires_grad_()
    t = (b * 10).double()
    ref = O.scl_logits_loss(x, t)
    ref.backward()
    rows, dx = ops.kl_logits((a * 10).cuda(), (b * 10).cuda(), want_grad=True, grad_scale=1.0 / a.numel())
    assert abs(rows.sum().item() / a.numel() - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
    assert (dx.cpu().double() - x.grad).abs().max() < 1e-8


@pytest.mark.gpu
def test_stage2_loss_matches_oracle_with_gradients():
    import slow_pace as S
    dev = torch.device("cuda:0")
    x = _inputs(B=16, C=11, d=64, seed=3)
    leaf = lambda t: t.clone().requires_grad_()
    img64, txt64 = leaf(x["img"]), leaf(x["txt"])
    lp64 = tuple(leaf(t) for t in x["lp"])
    ref, rterms, rcos = O.stage2_loss(img64, txt64, x["target"], x["zs_img"], x["zs_txt"], lp64, x["lp_img"], x["lp_txt"])
    ref.backward()
    head = S.Channel_LP(64, 11, device=dev)
    with torch.no_grad():
        head.scale1.copy_(x["lp"][0].float())
        head.bias1.copy_(x["lp"][1].float())
        head.fc.weight.copy_(x["lp"][2].float())
        head.fc.bias.copy_(x["lp"][3].float())
    D = lambda t: t.float().to(dev)
    img, txt = D(x["img"]).requires_grad_(), D(x["txt"]).requires_grad_()
    loss, terms, cos = S.stage2_loss(img, txt, x["target"].to(dev), D(x["zs_img"]), D(x["zs_txt"]), head, D(x["lp_img"]),
                                     D(x["lp_txt"]))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 2e-5 * abs(ref.item())
    for k, v in rterms.items():
        assert abs(terms[k].item() - v.item()) < 2e-5 * max(1.0, abs(v.item())), k
    assert (cos.detach().cpu().double() - rcos.detach()).abs().max() < 1e-3  # the north-star logit tolerance
    close = lambda got, want, tol: (got.detach().cpu().double() - want).abs().max().item() <= tol * max(want.abs().max().item(), 1e-12)
    assert close(img.grad, img64.grad, 2e-4) and close(txt.grad, txt64.grad, 2e-4)
    assert close(head.scale1.grad, lp64[0].grad, 2e-4) and close(head.bias1.grad, lp64[1].grad, 2e-4)
    assert close(head.fc.weight.grad, lp64[2].grad, 2e-4) and close(head.fc.bias.grad, lp64[3].grad, 2e-4)
    # with the MoCo branch: loss = the above + loss_aux (slow_pace.py:1677-1680,1688), adapter gradients from loss_aux only
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(16, 96, generator=g, dtype=torch.float64)
    ad = S.Moco_Adapter(96, 11, device=dev)
    ow = ad.fc.weight.detach().double().cpu().requires_grad_()
    ob = ad.fc.bias.detach().double().cpu().requires_grad_()
    aux = O.moco_aux_loss(feats, ow, ob, x["target"])
    aux.backward()
    loss2, terms2, _ = S.stage2_loss(D(x["img"]), D(x["txt"]), x["target"].to(dev), D(x["zs_img"]), D(x["zs_txt"]), head,
                                     D(x["lp_img"]), D(x["lp_txt"]), ad, D(feats))
    loss2.backward()
    assert abs(terms2["loss_aux"].item() - aux.item()) < 2e-5 * max(1.0, abs(aux.item()))
    assert abs(loss2.item() - (ref.item() + aux.item())) < 3e-5 * abs(ref.item() + aux.item())
    assert close(ad.fc.weight.grad, ow.grad, 2e-4) and close(ad.fc.bias.grad, ob.grad, 2e-4)


@pytest.mark.gpu
def test_stage2_trainer_step_matches_oracle(monkeypatch):
    """One Stage2Trainer step on a small CLIP with LoRA applied (frozen), 4 prompt tokens and 4 VPT tokens: loss and
    the gradients of ctx / VPT / head against the fp64 oracle composed from the same restated pieces (eval mode: no
    dropout, so the second no-grad image forward equals the first)."""
    import types
    import test_engine_gpu as T
    import lora_train_vlp as L
    import slow_pace as S
    from clipfs import synth
    dev = torch.device("cuda:0")
    cfg = synth.SMALL
    sd, model = T._build(cfg, dev, n_vpt=4)
    args = T._args("small", r=4)
    lw = synth.synth_lora(cfg, 4, seed=5)
    T._apply(model, cfg, args, lw, monkeypatch)
    model.eval()
    B, C, d = 6, 5, cfg.embed_dim
    g = torch.Generator().manual_seed(1)
    unit = lambda t: t / t.norm(dim=-1, keepdim=True)
    images = synth.synth_images(B, cfg.image_resolution, seed=3)
    target = synth.synth_labels(B, C, seed=2)
    index = torch.arange(B)
    zs_img = unit(torch.randn(B, d, generator=g, dtype=torch.float64))
    zs_txt = unit(torch.randn(C, d, generator=g, dtype=torch.float64))
    # prompts "<SOT> ctx ctx ctx ctx w1 w2 . <EOT>" as token ids (the tokenizer needs no special vocabulary here)
    ids = synth.synth_captions(C, cfg.context_length, cfg.vocab_size, seed=4, max_len=12)
    ctx0 = model.token_embedding.weight.data[ids[0, 1:5].to(dev)].clone()
    learner = S.VLPromptLearner.__new__(S.VLPromptLearner)
    torch.nn.Module.__init__(learner)
    learner.ctx = torch.nn.Parameter(ctx0.clone())
    learner.tokenized_prompts, learner.n_ctx, learner.n_cls = ids.to(dev), 4, C
    learner._model = [model]
    head = S.Channel_LP(d, C, device=dev)
    with torch.no_grad():
        head.fc.weight.copy_(zs_txt.float())
        head.scale1.copy_(1 + 0.1 * torch.randn(d, generator=g))
    tr = S.Stage2Trainer(model, learner, head, zs_img, zs_txt, lr=1e-3, total_epoch=20)
    p0 = [p.detach().clone() for p in tr.params]
    loss, terms, cos = tr.step(images.to(dev), target.to(dev), index)
    # ---- oracle
    sd64 = {k: v.double() for k, v in sd.items()}
    tl, vl = T._oracle_lora(lw, cfg)
    s = O.lora_scaling(1, 4)
    ctx64 = p0[0].double().cpu().requires_grad_()
    vpt64 = p0[1].double().cpu().requires_grad_()
    emb = sd64["token_embedding.weight"][ids]
    prompts = torch.cat([emb[:, :1], ctx64.unsqueeze(0).expand(C, -1, -1), emb[:, 5:]], dim=1)   # slow_pace.py:185-199
    txt = O.encode_text(sd64, ids, tl, s, embeds=prompts)
    img = O.encode_image(sd64, images.double(), vl, s, vpt=vpt64)
    lp64 = [t.double().cpu().requires_grad_() for t in p0[2:]]
    ref, rterms, _ = O.stage2_loss(img, txt, target, zs_img, zs_txt, tuple(lp64), img.detach(), zs_txt)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 5e-5 * abs(ref.item())
    got = [p.grad for p in tr.params]
    want = [ctx64.grad, vpt64.grad] + [t.grad for t in lp64]
    for gq, w in zip(got, want):
        assert (gq.detach().cpu().double() - w).abs().max().item() <= 5e-4 * max(w.abs().max().item(), 1e-6)
    # AdamW with the pre-step learning rate, then the cosine schedule advanced once (:1696-1697)
    assert abs(tr.lr - O.cosine_annealing_lr(1e-3, 1, 20)) < 1e-12
    for p, q, w in zip(tr.params, p0, want):
        q64 = q.double().cpu()
        exp, _, _ = O.jt_adamw_step(q64, w, torch.zeros_like(q64), torch.zeros_like(q64), 1, lr=1e-3, weight_decay=1e-2)
        assert (p.detach().cpu().double() - exp).abs().max().item() < 1e-6


@pytest.mark.gpu
def test_load_lora_swa_averages_saved_files(monkeypatch, tmp_path):
    """slow_pace.py:736-816: three save_lora files -> their element-wise mean lands in the adapters; sub-folders are
    skipped; metadata mismatch / missing folder raise like the reference."""
    import numpy as np
    import test_engine_gpu as T
    import lora_train_vlp as L
    import slow_pace as S
    from clipfs import synth
    dev = torch.device("cuda:0")
    cfg = synth.TINY
    _, model = T._build(cfg, dev)
    args = T._args("tiny", r=4)
    folder = tmp_path / "swa"
    (folder / "nested").mkdir(parents=True)
    sets = [synth.synth_lora(cfg, 4, seed=s) for s in (21, 22, 23)]
    layers = T._apply(model, cfg, args, sets[0], monkeypatch)
    names = {"q": "q_proj", "k": "k_proj", "v": "v_proj"}
    for n, lw in enumerate(sets):
        with torch.no_grad():
            for i, layer in enumerate(layers):
                for p in args.params:
                    m = getattr(layer, names[p])
                    m.w_lora_A.copy_(torch.from_numpy(lw[f"layer_{i}"][names[p]]["w_lora_A"]))
                    m.w_lora_B.copy_(torch.from_numpy(lw[f"layer_{i}"][names[p]]["w_lora_B"]))
        L.save_lora(args, n, layers, save_path=str(folder / f"lora_{n}.pkl"))
    assert S.load_lora_swa(args, layers, str(folder)) == 3
    for i, layer in enumerate(layers):
        for p in args.params:
            for k in ("w_lora_A", "w_lora_B"):
                want = np.mean([np.asarray(lw[f"layer_{i}"][names[p]][k], np.float64) for lw in sets], axis=0)
                got = getattr(getattr(layer, names[p]), k).detach().cpu().double().numpy()
                assert np.abs(got - want).max() < 1e-7
    with pytest.raises(ValueError):
        S.load_lora_swa(T._args("tiny", r=8), layers, str(folder))
    with pytest.raises(FileNotFoundError):
        S.load_lora_swa(args, layers, str(folder) + ".missing")


@pytest.mark.gpu
def test_pre_load_zs_and_prompt_queue(monkeypatch):
    """slow_pace.py:1435-1454: cached zero-shot MTA features = per image solve_mta(mode) over its views (oracle per
    image); PromptQueue (README.md:22): fixed-length queue, blend = normalise(w mean(queue) + (1 - w) hand)."""
    import test_engine_gpu as T
    import slow_pace as S
    from clipfs import synth
    dev = torch.device("cuda:0")
    cfg = synth.SMALL
    sd, model = T._build(cfg, dev)
    model.eval()
    n_img, V, C, d = 2, 9, 7, cfg.embed_dim
    g = torch.Generator().manual_seed(5)
    R = cfg.image_resolution
    base = torch.randn(n_img, 1, 3, R, R, generator=g)
    views = (base + 0.3 * torch.randn(n_img, V, 3, R, R, generator=g)).contiguous()
    text = O.l2_normalize(torch.randn(C, d, generator=g, dtype=torch.float64)).float()
    feats = S.pre_load_zs(model, views.to(dev), text.to(dev))
    sd64 = {k: v.double() for k, v in sd.items()}
    assert feats.shape == (n_img, d)
    for i in range(n_img):
        f = O.l2_normalize(O.encode_image(sd64, views[i].double())).float()
        want = O.solve_mta(f, text.t(), return_mode=True)
        assert (feats[i:i + 1].cpu().double() - want.double()).abs().max() < 5e-5
    q = S.PromptQueue(maxlen=2)
    hand = torch.randn(C, d, generator=g)
    assert torch.allclose(q.blend(hand.to(dev)).cpu(), O.l2_normalize(hand.double()).float(), atol=1e-6)
    fs = [torch.randn(C, d, generator=g) for _ in range(3)]
    for f in fs:
        q.push(f.to(dev))
    assert len(q) == 2  # the oldest entry dropped out
    mean = torch.stack([O.l2_normalize(f.double()) for f in fs[1:]]).mean(0)
    want = O.l2_normalize(0.25 * mean + 0.75 * O.l2_normalize(hand.double()))
    assert (q.blend(hand.to(dev), weight=0.25).cpu().double() - want).abs().max() < 1e-6
