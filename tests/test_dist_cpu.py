"""N > 1 path on CPU: two gloo ranks run the SAME exchange pattern as LoRATrainer (clipfs/dist.py helpers:
image shard + B_local/B_global loss scaling, class-sharded text tower with all-gather of the class features and
a summed gradient, ONE all-reduce of the flat gradient) with the oracle as the compute, and must reproduce
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


def _setup():
    from clipfs import synth
    cfg = synth.TINY
    sd = {k: v.double() for k, v in synth.synth_state_dict(cfg, seed=21, perturb=True).items()}
    lw = synth.synth_lora(cfg, 4, seed=22)
    B, Cn = 6, 7
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


def _rank_main(rank, world, port, shard_text, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from clipfs import dist as D
    from oracle import clip_oracle as O
    cfg, sd, lw, img, cap, tgt = _setup()
    tl, vl, flat = _adapters(cfg, lw)
    ctx = sd["token_embedding.weight"][[9, 10, 11, 12]].clone().requires_grad_()
    B, Cn = img.shape[0], cap.shape[0]
    assert D.world_info() == (rank, world)
    lo, hi = D.shard_bounds(B, rank, world)
    c_lo, c_hi = D.shard_bounds(Cn, rank, world) if shard_text else (0, Cn)
    # text: my classes only, with grad
    pe = O.build_prompts(ctx, sd["token_embedding.weight"], cap[c_lo:c_hi])
    emb = O.encode_text(sd, cap[c_lo:c_hi], tl, 0.5, embeds=pe)
    txt_local = O.class_text_features(emb, list(range(c_hi - c_lo)), c_hi - c_lo).t()  # [c, d]
    if shard_text:
        txt_full = D.allgather_rows(txt_local.detach(), c_lo, c_hi, Cn, cfg.embed_dim, txt_local.detach())
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
        D.allreduce_sum_(d_txt)
    txt_local.backward(d_txt[c_lo:c_hi] if shard_text else d_txt)
    grads = torch.cat([(t.grad if t.grad is not None else torch.zeros_like(t)).reshape(-1) for t in flat] +
                      [ctx.grad.reshape(-1)])
    D.allreduce_sum_(grads)  # the single flat all-reduce
    total = torch.tensor([loss_local.item()], dtype=torch.float64)
    D.allreduce_sum_(total)
    if rank == 0:
        np.savez(os.path.join(out_dir, f"dp_{int(shard_text)}.npz"), grads=grads.numpy(), loss=total.numpy())
    dist.destroy_process_group()


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
