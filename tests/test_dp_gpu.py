"""The data-parallel trainer on the GPU: two ranks share the one MI355X of the test box (gloo moves the
CUDA tensors; on an 8-GPU node the same code runs over RCCL), each takes half of the image batch and -- with
shard_text -- half of the classes; the summed flat gradient and the AdamW-updated parameters must equal a
single-process step on the whole batch."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(dev, p=0.0):
    import lora_train_vlp as L
    from clipfs import synth
    from jclip.model import build_model
    cfg = synth.SMALL
    sd = synth.synth_state_dict(cfg, seed=11, perturb=True)
    model = build_model(sd, device=dev)
    args = types.SimpleNamespace(encoder="both", position="all", backbone="small", params=["q", "k", "v"], r=4, alpha=1,
                                 dropout_rate=p)
    saved = L.INDEX_POSITIONS_TEXT["all"]
    L.INDEX_POSITIONS_TEXT["all"] = list(range(cfg.transformer_layers))
    L.INDEX_POSITIONS_VISION["small"] = {"all": list(range(cfg.vision_layers))}
    try:
        layers = L.apply_lora(args, model)
    finally:
        L.INDEX_POSITIONS_TEXT["all"] = saved
        del L.INDEX_POSITIONS_VISION["small"]
    lw = synth.synth_lora(cfg, 4, seed=5)
    names = {"q": "q_proj", "k": "k_proj", "v": "v_proj"}
    with torch.no_grad():
        for i, layer in enumerate(layers):
            for pr in "qkv":
                m = getattr(layer, names[pr])
                m.w_lora_A.copy_(torch.from_numpy(lw[f"layer_{i}"][names[pr]]["w_lora_A"]))
                m.w_lora_B.copy_(torch.from_numpy(lw[f"layer_{i}"][names[pr]]["w_lora_B"]))
    ctx = torch.nn.Parameter(sd["token_embedding.weight"][[5, 6, 7, 8]].clone().to(dev))
    B, Cn = 8, 9
    img = synth.synth_images(B, cfg.image_resolution, seed=3).to(dev)
    cap = synth.synth_captions(Cn, cfg.context_length, cfg.vocab_size, seed=4, max_len=12).to(dev)
    tgt = synth.synth_labels(B, Cn, seed=2).to(dev)
    return L, model, ctx, img, cap, tgt


def _rank_main(rank, world, port, shard_text, out_dir, p=0.0, backend="gloo", force=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in (os.path.join(root, "jittor-clip-fewshot_amd"), root):
        if path not in sys.path:
            sys.path.insert(0, path)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from clipfs import dist as D
    dev = torch.device("cuda:0")
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    D.FORCE_COLLECTIVES = force
    L, model, ctx, img, cap, tgt = _make(dev, p)
    model.train(p > 0)
    tr = L.LoRATrainer(model, prompt_ctx=ctx, shard_text=shard_text)
    assert (tr.rank, tr.world) == (rank, world)
    assert tr.collectives_per_step == (3 if shard_text else 1)
    lo, hi = D.shard_bounds(img.shape[0], rank, world)
    tr.flat.zero_grad()
    tr.time_collectives = True
    loss_sum, _, _ = tr.forward_backward(img[lo:hi].contiguous(), cap, tgt[lo:hi].contiguous(), 1, img.shape[0],
                                         row_offset=lo)
    tr.optimizer_step()
    torch.cuda.synchronize()
    times = tr.collective_times_ms()
    assert sorted(times) == (["all_gather", "all_reduce", "reduce_scatter"] if shard_text else ["all_reduce"]), times
    total = loss_sum.clone()
    D.allreduce_sum_(total)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), grads=tr.flat.grads.cpu().numpy(), params=tr.flat.params.cpu().numpy(),
                 loss=total.cpu().numpy())
    dist.destroy_process_group()


def _single(dev, p):
    L, model, ctx, img, cap, tgt = _make(dev, p)
    model.train(p > 0)
    tr = L.LoRATrainer(model, prompt_ctx=ctx)
    tr.flat.zero_grad()
    loss_sum, _, _ = tr.forward_backward(img, cap, tgt)
    tr.optimizer_step()
    return tr.flat.grads.cpu().numpy(), tr.flat.params.cpu().numpy(), loss_sum.item()


def _compare(tmp_path, want_g, want_p, want_loss):
    z = np.load(os.path.join(str(tmp_path), "dp.npz"))
    scale = np.abs(want_g).max()
    assert scale > 1e-5
    assert np.abs(z["grads"] - want_g).max() < 2e-5 * scale + 1e-9
    assert np.abs(z["params"] - want_p).max() < 1e-6
    assert abs(float(z["loss"][0]) - want_loss) < 1e-4


def test_two_ranks_with_dropout_draw_the_single_process_masks(tmp_path):
    """LoRA dropout 0.25 under data parallelism: the Philox counter is the GLOBAL row (image row_offset, class-block
    offset), so a 2-rank step reproduces the one-process step bit-for-bit in its masks -- the gradients agree to
    rounding, and ranks do not share masks."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    want_g, want_p, want_loss = _single(torch.device("cuda:0"), 0.25)
    mp.spawn(_rank_main, args=(2, _free_port(), True, str(tmp_path), 0.25), nprocs=2, join=True)
    _compare(tmp_path, want_g, want_p, want_loss)


def test_rccl_entry_points_single_rank(tmp_path):
    """The three collectives of the sharded step issued through RCCL itself (backend "nccl", one rank on the one GPU of
    the test box -- RCCL refuses two ranks on one device, and the 8-GPU run belongs to the driver): checks that
    all_gather_into_tensor / reduce_scatter_tensor / all_reduce accept the trainer's buffers and leave the step's
    result unchanged."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    want_g, want_p, want_loss = _single(torch.device("cuda:0"), 0.0)
    mp.spawn(_rank_main, args=(1, _free_port(), True, str(tmp_path), 0.0, "nccl", True), nprocs=1, join=True)
    _compare(tmp_path, want_g, want_p, want_loss)


@pytest.mark.parametrize("shard_text", [True, False])
def test_two_ranks_match_single_process(tmp_path, shard_text):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    dev = torch.device("cuda:0")
    L, model, ctx, img, cap, tgt = _make(dev)
    model.eval()
    tr = L.LoRATrainer(model, prompt_ctx=ctx)
    tr.flat.zero_grad()
    loss_sum, _, _ = tr.forward_backward(img, cap, tgt)
    tr.optimizer_step()
    want_g, want_p = tr.flat.grads.cpu().numpy(), tr.flat.params.cpu().numpy()
    mp.spawn(_rank_main, args=(2, _free_port(), shard_text, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(str(tmp_path), "dp.npz"))
    scale = np.abs(want_g).max()
    assert scale > 1e-5
    assert np.abs(z["grads"] - want_g).max() < 2e-5 * scale + 1e-9
    assert np.abs(z["params"] - want_p).max() < 1e-6
    assert abs(float(z["loss"][0]) - loss_sum.item()) < 1e-4
