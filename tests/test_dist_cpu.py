"""N > 1 path on CPU: two gloo ranks run the SAME exchange pattern as LoRATrainer (clipfs/dist.py helpers:
image shard + B_local/B_global loss scaling, class-sharded text tower with ONE all_gather_into_tensor of the class-feature
blocks and ONE reduce_scatter_tensor of their gradient, ONE all-reduce of the flat gradient) with the oracle as the compute, and must reproduce
the single-process gradients of the full batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(B=6, Cn=7):
    from clipfs import synth
    cfg = synth.TINY
    sd = {k: v.double() for k, v in synth.synth_state_dict(cfg, seed=21, perturb=True).items()}
    lw = synth.synth_lora(cfg, 4, seed=22)
    img = synth.synth_images(B, cfg.image_resolution, seed=23).double()
    cap = synth.synth_captions(Cn, cfg.context_length, cfg.vocab_size, seed=24, max_len=9)
    tgt = synth.synth_labels(B, Cn, seed=25)
    return cfg, sd, lw, img, cap, tgt


def _adapters(cfg, lw):
    nt = cfg.transformer_layers
    conv = lambda d: {p: {k: torch.from_numpy(v).double().requires_grad_() for k, v in ab.items()} for p, ab in d.items()}
    tl = {b: conv(lw[f"layer_{b}"]) for b in range(nt)}
    vl = {b: conv(lw[f"layer_{nt + b}"]) for b in range(cfg.vision_layers)}
    flat = [t for blk in list(tl.values()) + list(vl.values()) for ab in blk.values() for t in ab.values()]
    return tl, vl, flat


def _rank_main(rank, world, port, shard_text, out_dir, B=6, Cn=7):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2 if world <= 2 else 1)
    from clipfs import dist as D
    from oracle import clip_oracle as O
    cfg, sd, lw, img, cap, tgt = _setup(B, Cn)
    tl, vl, flat = _adapters(cfg, lw)
    ctx = sd["token_embedding.weight"][[9, 10, 11, 12]].clone().requires_grad_()
    B, Cn = img.shape[0], cap.shape[0]
    assert D.world_info() == (rank, world)
    lo, hi = D.shard_bounds(B, rank, world)
    c_lo, c_hi = D.block_bounds(Cn, rank, world) if shard_text else (0, Cn)
    S = D.block_rows(Cn, world)
    # text: my classes only, with grad
    pe = O.build_prompts(ctx, sd["token_embedding.weight"], cap[c_lo:c_hi])
    emb = O.encode_text(sd, cap[c_lo:c_hi], tl, 0.5, embeds=pe)
    txt_local = O.class_text_features(emb, list(range(c_hi - c_lo)), c_hi - c_lo).t()  # [c, d]
    if shard_text:
        send = torch.zeros(S, cfg.embed_dim, dtype=torch.float64)
        send[:c_hi - c_lo] = txt_local.detach()
        txt_full = D.allgather_blocks(send)[:Cn]
    else:
        txt_full = txt_local.detach()
    txt_leaf = txt_full.clone().requires_grad_()
    # images: my shard, loss pre-scaled by B_local / B_global
    fi = O.encode_image(sd, img[lo:hi], vl, 0.5)
    logits = O.train_logits(fi, txt_leaf.t())
    loss_local = O.jt_cross_entropy(logits, tgt[lo:hi]) * ((hi - lo) / B)
    loss_local.backward()
    d_txt = txt_leaf.grad.clone()
    if shard_text:
        dfull = torch.zeros(world * S, cfg.embed_dim, dtype=torch.float64)
        dfull[:Cn] = d_txt
        txt_local.backward(D.reduce_scatter_blocks(dfull)[:c_hi - c_lo])
    else:
        txt_local.backward(d_txt)
    grads = torch.cat([(t.grad if t.grad is not None else torch.zeros_like(t)).reshape(-1) for t in flat] +
                      [ctx.grad.reshape(-1)])
    D.allreduce_sum_(grads)  # the single flat all-reduce
    total = torch.tensor([loss_local.item()], dtype=torch.float64)
    D.allreduce_sum_(total)
    if rank == 0:
        np.savez(os.path.join(out_dir, f"dp_{int(shard_text)}.npz"), grads=grads.numpy(), loss=total.numpy())
    dist.destroy_process_group()


def test_eight_rank_gloo_at_cfg3_counts(tmp_path):
    """cfg-3's partition on 8 ranks (gloo, CPU, oracle compute on the TINY model): B = 256 -> 32 images per rank,
    C = 403 classes -> blocks of S = 51 with a short last block (46 rows), one all_gather + one reduce_scatter + one
    all-reduce; the summed gradient must equal the single-process gradient of the whole batch."""
    from clipfs import dist as D
    from oracle import clip_oracle as O
    B, Cn, world = 256, 403, 8
    assert D.block_rows(Cn, world) == 51 and D.block_bounds(Cn, 7, world) == (357, 403)
    assert [D.shard_bounds(B, r, world) for r in (0, 7)] == [(0, 32), (224, 256)]
    cfg, sd, lw, img, cap, tgt = _setup(B, Cn)
    tl, vl, flat = _adapters(cfg, lw)
    ctx = sd["token_embedding.weight"][[9, 10, 11, 12]].clone().requires_grad_()
    loss, _ = O.train_step_loss(sd, img, cap, tgt, tl, vl, 0.5, ctx=ctx)
    loss.backward()
    want = torch.cat([t.grad.reshape(-1) for t in flat] + [ctx.grad.reshape(-1)]).numpy()
    mp.spawn(_rank_main, args=(world, _free_port(), True, str(tmp_path), B, Cn), nprocs=world, join=True)
    z = np.load(os.path.join(str(tmp_path), "dp_1.npz"))
    assert abs(float(z["loss"][0]) - loss.item()) < 1e-11
    assert np.allclose(z["grads"], want, atol=1e-11, rtol=1e-8)
    assert np.abs(want).max() > 1e-5


def test_exchange_helpers_reject_wrong_shapes():
    from clipfs import dist as D
    blk = torch.zeros(3, 4)
    with pytest.raises(ValueError):
        D.allgather_blocks(blk, torch.zeros(4, 4))
    with pytest.raises(ValueError):
        D.allgather_blocks(torch.zeros(4, 6)[:, :4], torch.zeros(4, 4))
    with pytest.raises(ValueError):
        D.reduce_scatter_blocks(torch.zeros(3, 4), torch.zeros(2, 4))
    assert D.allgather_blocks(blk).shape == (3, 4) and D.reduce_scatter_blocks(blk).shape == (3, 4)


@pytest.mark.parametrize("shard_text", [True, False])
def test_two_rank_gloo_matches_single_process(tmp_path, shard_text):
    from oracle import clip_oracle as O
    cfg, sd, lw, img, cap, tgt = _setup()
    tl, vl, flat = _adapters(cfg, lw)
    ctx = sd["token_embedding.weight"][[9, 10, 11, 12]].clone().requires_grad_()
    loss, _ = O.train_step_loss(sd, img, cap, tgt, tl, vl, 0.5, ctx=ctx)
    loss.backward()
    want = torch.cat([t.grad.reshape(-1) for t in flat] + [ctx.grad.reshape(-1)]).numpy()
    mp.spawn(_rank_main, args=(2, _free_port(), shard_text, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(str(tmp_path), f"dp_{int(shard_text)}.npz"))
    assert abs(float(z["loss"][0]) - loss.item()) < 1e-12
    assert np.allclose(z["grads"], want, atol=1e-12, rtol=1e-9)
    assert np.abs(want).max() > 1e-4


# ---- cfg-4 (MTA TTA / OOD scoring) on N ranks: images sharded, integer results gathered -------------------------

def _tta_inputs():
    g = torch.Generator().manual_seed(31)
    n_img, V, C, d = 7, 9, 380, 32   # 7 images over 3 ranks: shards of 2 / 2 / 3; C > 373 so both base and new occur
    from oracle import clip_oracle as O
    text = O.l2_normalize(torch.randn(C, d, generator=g, dtype=torch.float64))
    feats = torch.randn(n_img, V, d, generator=g, dtype=torch.float64)
    for i, c in ((1, 375), (4, 379), (6, 376), (0, 10), (3, 200)):  # views clustered around a class prototype
        feats[i] = text[c] + 0.15 * feats[i]
    return O.l2_normalize(feats), text


def _oracle_scores(feats, text, lo, hi):
    from oracle import clip_oracle as O
    top5, base = [], []
    for i in range(lo, hi):
        logits = O.solve_mta(feats[i], text.t())
        top5.append(O.jt_topk(logits, 5).reshape(1, 5))
        base.append(bool(O.ood_is_base(logits)[0]))
    return torch.cat(top5).to(torch.int32), torch.tensor(base)


def _tta_rank_main(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import ood
    feats, text = _tta_inputs()
    calls = []

    def score_fn(lo, hi):
        calls.append((lo, hi))
        return _oracle_scores(feats, text, lo, hi)

    top5, is_base = ood.score_images_sharded(score_fn, feats.shape[0], torch.device("cpu"))
    np.savez(os.path.join(out_dir, f"tta_{rank}.npz"), top5=top5.numpy(), base=is_base.numpy(), calls=np.array(calls))
    dist.destroy_process_group()


def test_tta_scoring_sharded_over_three_gloo_ranks(tmp_path):
    """rank r scores its contiguous shard of the source images (all views of an image stay on
    one rank), one integer exchange assembles top-5 labels and base/new flags on every rank; must equal the
    single-process result, on every rank, with uneven shards."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jittor-clip-fewshot_amd"))
    feats, text = _tta_inputs()
    want_top5, want_base = _oracle_scores(feats, text, 0, feats.shape[0])
    assert want_base.any() and not want_base.all()
    mp.spawn(_tta_rank_main, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    seen = []
    for r in range(3):
        z = np.load(os.path.join(str(tmp_path), f"tta_{r}.npz"))
        assert np.array_equal(z["top5"], want_top5.numpy()) and np.array_equal(z["base"], want_base.numpy())
        seen += [tuple(c) for c in z["calls"]]
    assert sorted(seen) == [(0, 2), (2, 4), (4, 7)]  # disjoint cover of the 7 images
