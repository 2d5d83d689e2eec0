"""Host runtime of the HIP engine: collects device pointers from the ``jclip`` model objects into the
C descriptors of include/clipfs.h, owns activation workspaces, and runs the towers forward/backward
with ONE C call per pass (csrc/tower.hip sequences the kernels).  PyTorch supplies device memory,
streams and ``torch.distributed`` only.

Replaces, for the hot path: VisionTransformer.execute / CLIP.encode_text / CLIP.execute
(jclip/model.py:104-126,202-232), TextEncoder.execute (slow_pace.py:837-848), the step body of
run_lora (lora_train_vlp.py:956-1002) and Jittor's autograd + AdamW behind it."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, ops
from ._lib import Block, Tower, check


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _mix_seed(base: int, step: int) -> int:
    """64-bit non-zero dropout seed for (base, step) -- splitmix64 finaliser."""
    z = (base * 0x9E3779B97F4A7C15 + step * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 31
    return z or 1


class _TowerRT:
    """One transformer tower: pointer collection + workspace cache."""

    def __init__(self, transformer, seq: int, stream0: int):
        self.mod = transformer
        self.seq = seq
        self.width = transformer.width
        self.heads = transformer.heads
        self.layers = transformer.layers
        self.causal = bool(transformer.causal)
        self.stream0 = stream0
        self._bufs: Dict[Tuple[str, int], torch.Tensor] = {}
        self._wt: Dict[int, Dict[str, torch.Tensor]] = {}
        self._planes: Dict[Tuple[int, str, str], torch.Tensor] = {}
        self.precision = "fp32"  # "fp32": exact v_mfma_f32_32x32x2_f32 | "bf16x3": split-bf16, 3 MFMA products
        self.lora_r = 0
        self.lora_scale = 0.0
        self.lora_dropout = 0.0

    # -- descriptor ------------------------------------------------------------------------------
    def _transposed(self, i: int, blk) -> Dict[str, torch.Tensor]:
        wt = self._wt.get(i)
        if wt is None:
            a = blk.attn
            w_qkv = a.qkv_weight if getattr(a, "is_lora_mha", False) else a.in_proj_weight
            w_o = a.proj.weight if getattr(a, "is_lora_mha", False) else a.out_proj.weight
            wt = {"qkv": w_qkv.data.t().contiguous(), "o": w_o.data.t().contiguous(),
                  "fc": blk.mlp.c_fc.weight.data.t().contiguous(), "pr": blk.mlp.c_proj.weight.data.t().contiguous()}
            self._wt[i] = wt
        return wt

    def _plane(self, i: int, name: str, w: torch.Tensor) -> torch.Tensor:
        key = (i, name, self.precision)
        pl = self._planes.get(key)
        if pl is None:
            pl = ops.split_bf16(w.contiguous()) if self.precision == "bf16x3" else ops.to_f16(w.contiguous())
            self._planes[key] = pl
        return pl

    def descriptor(self, train: bool, seed: int, seq: Optional[int] = None, row0: int = 0) -> Tower:
        blocks = (Block * self.layers)()
        r, scale, p = 0, 0.0, 0.0
        for i, blk in enumerate(self.mod.resblocks):
            a = blk.attn
            b = blocks[i]
            lora = getattr(a, "is_lora_mha", False)
            if lora:
                w_qkv, b_qkv, w_o, b_o = a.qkv_weight, a.qkv_bias, a.proj.weight, a.proj.bias
            else:
                w_qkv, b_qkv, w_o, b_o = a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias
            b.ln1_g, b.ln1_b = _ptr(blk.ln_1.weight), _ptr(blk.ln_1.bias)
            b.ln2_g, b.ln2_b = _ptr(blk.ln_2.weight), _ptr(blk.ln_2.bias)
            b.w_qkv, b.b_qkv = _ptr(w_qkv), _ptr(b_qkv)
            b.w_o, b.b_o = _ptr(w_o), _ptr(b_o)
            b.w_fc, b.b_fc = _ptr(blk.mlp.c_fc.weight), _ptr(blk.mlp.c_fc.bias)
            b.w_pr, b.b_pr = _ptr(blk.mlp.c_proj.weight), _ptr(blk.mlp.c_proj.bias)
            if train:
                wt = self._transposed(i, blk)
                b.w_qkv_t, b.w_o_t, b.w_fc_t, b.w_pr_t = _ptr(wt["qkv"]), _ptr(wt["o"]), _ptr(wt["fc"]), _ptr(wt["pr"])
            if self.precision != "fp32":
                b.w_qkv_p = _ptr(self._plane(i, "qkv", w_qkv.data))
                b.w_o_p = _ptr(self._plane(i, "o", w_o.data))
                b.w_fc_p = _ptr(self._plane(i, "fc", blk.mlp.c_fc.weight.data))
                b.w_pr_p = _ptr(self._plane(i, "pr", blk.mlp.c_proj.weight.data))
                if train:
                    b.w_qkv_t_p = _ptr(self._plane(i, "qkv_t", wt["qkv"]))
                    b.w_o_t_p = _ptr(self._plane(i, "o_t", wt["o"]))
                    b.w_fc_t_p = _ptr(self._plane(i, "fc_t", wt["fc"]))
                    b.w_pr_t_p = _ptr(self._plane(i, "pr_t", wt["pr"]))
            b.lora_mask = 0
            if lora and a.r > 0:
                if r and (a.r != r or abs(a.scaling - scale) > 0 or abs(a.dropout_rate - p) > 0):
                    raise ValueError("all LoRA layers of one tower must share r / alpha / dropout")
                r, scale, p = a.r, float(a.scaling), float(a.dropout_rate)
                b.lora_mask = a.lora_mask
                b.lora_a_qkv, b.lora_b_qkv = _ptr(a.lora_A_qkv), _ptr(a.lora_B_qkv)
                b.g_lora_a_qkv, b.g_lora_b_qkv = _ptr(a.grad_A_qkv), _ptr(a.grad_B_qkv)
                if a.lora_mask & 8:
                    b.lora_a_o, b.lora_b_o = _ptr(a.lora_A_o), _ptr(a.lora_B_o)
                    b.g_lora_a_o, b.g_lora_b_o = _ptr(a.grad_A_o), _ptr(a.grad_B_o)
        t = _lib.new_tower()
        t.width, t.heads, t.layers, t.seq, t.causal = (self.width, self.heads, self.layers, seq or self.seq,
                                                       int(self.causal))
        t.lora_r, t.lora_scale, t.lora_dropout = r, scale, p
        # dropout follows the MODULE's train mode (LinearLoRA.execute, lora_train_vlp.py:297-298), not whether
        # activations are saved: a no-grad forward in train mode (slow_pace.py:1659-1661) still drops
        t.dropout_seed = seed if p > 0 else 0
        t.dropout_stream0 = self.stream0
        t.dropout_row0 = row0 * (seq or self.seq)  # first TOKEN row of this call in the global batch
        t.weight_format = {"fp32": 0, "bf16x3": 1, "fp16": 2}[self.precision]
        t.blocks = C.cast(blocks, C.POINTER(Block))
        t._blocks_keepalive = blocks  # ctypes array must outlive the call
        self.lora_r, self.lora_scale, self.lora_dropout = r, scale, p
        return t

    def lora_dropout_rate(self) -> float:
        return max((float(b.attn.dropout_rate) for b in self.mod.resblocks
                    if getattr(b.attn, "is_lora_mha", False) and b.attn.r > 0), default=0.0)

    def has_lora(self) -> bool:
        return any(getattr(b.attn, "is_lora_mha", False) and b.attn.r > 0 for b in self.mod.resblocks)

    # -- workspaces -----------------------------------------------------------------------------
    def buffer(self, kind: str, n_floats: int, device) -> torch.Tensor:
        key = (kind, n_floats)
        buf = self._bufs.get(key)
        if buf is None:
            # drop stale sizes of the same kind before allocating (batch size changed)
            for k in [k for k in self._bufs if k[0] == kind]:
                del self._bufs[k]
            buf = torch.empty(max(n_floats, 4), device=device, dtype=torch.float32)
            self._bufs[key] = buf
        return buf

    def _attach_counters(self, t: Tower, batch: int, device) -> None:
        """Stream-K arrival counters of the tower's GEMMs: zeroed once here, left zero by every launch; one buffer per
        tower (the two towers run on different streams)."""
        n = _lib.load().clipfs_tower_counter_ints(C.byref(t), batch)
        if n == 0:
            return
        buf = self._bufs.get(("counters", 0))
        if buf is None or buf.numel() < n:
            buf = torch.zeros(n, device=device, dtype=torch.int32)
            self._bufs[("counters", 0)] = buf
        t.gemm_counters, t.gemm_counters_ints = buf.data_ptr(), buf.numel()

    def forward(self, x: torch.Tensor, batch: int, train: bool, seed: int, seq: Optional[int] = None,
                own_saved: bool = False, row0: int = 0, rows: Optional[torch.Tensor] = None):
        """``train`` = keep the activations the backward needs.  ``own_saved``: give this call its OWN saved-activation
        tensor (the autograd route: several grad-enabled forwards of one tower may precede one backward, e.g. the 13
        caption chunks of encode_text_in_batches, lora_train_vlp.py:905-912); otherwise the per-tower cached buffer
        is reused (LoRATrainer: exactly one forward per backward)."""
        lib = _lib.load()
        t = self.descriptor(train, seed, seq, row0)
        self._attach_counters(t, batch, x.device)
        scratch = self.buffer("scratch", lib.clipfs_tower_scratch_floats(C.byref(t), batch), x.device)
        saved = None
        if train:
            n = lib.clipfs_tower_saved_floats(C.byref(t), batch)
            saved = torch.empty(max(n, 4), device=x.device, dtype=torch.float32) if own_saved else \
                self.buffer("saved", n, x.device)
        if rows is not None:
            # the caller reads one row per sequence (``rows`` [batch] int32): the last block's output projection and MLP
            # run on those rows only; the matching backward is ``backward_sparse`` with the same rows
            assert rows.dtype == torch.int32 and rows.is_cuda and rows.numel() >= batch
            check(lib.clipfs_tower_fwd_rows(C.byref(t), x.data_ptr(), rows.data_ptr(), batch, _ptr(saved), scratch.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "tower_fwd_rows")
        else:
            check(lib.clipfs_tower_fwd(C.byref(t), x.data_ptr(), batch, _ptr(saved), scratch.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "tower_fwd")
        return saved

    def backward(self, dx: torch.Tensor, batch: int, saved: torch.Tensor, seed: int, stop_at_input: bool,
                 seq: Optional[int] = None, row0: int = 0):
        lib = _lib.load()
        t = self.descriptor(True, seed, seq, row0)
        self._attach_counters(t, batch, dx.device)
        scratch = self.buffer("scratch", lib.clipfs_tower_scratch_floats(C.byref(t), batch), dx.device)
        check(lib.clipfs_tower_bwd(C.byref(t), dx.data_ptr(), batch, saved.data_ptr(), scratch.data_ptr(),
                                   int(stop_at_input), torch.cuda.current_stream().cuda_stream), "tower_bwd")


    def backward_sparse(self, dxs: torch.Tensor, rows: torch.Tensor, batch: int, saved: torch.Tensor, seed: int,
                        stop_at_input: bool, seq: Optional[int] = None, row0: int = 0) -> torch.Tensor:
        """Backward from a gradient that is non-zero in ONE row per sequence (``dxs`` [batch, width] at token ``rows``
        [batch] int32): the class token of the image head, the EOT token of the text head.  Returns the dense gradient
        wrt the tower input (uninitialised when ``stop_at_input``)."""
        lib = _lib.load()
        t = self.descriptor(True, seed, seq, row0)
        self._attach_counters(t, batch, dxs.device)
        scratch = self.buffer("scratch", lib.clipfs_tower_scratch_floats(C.byref(t), batch), dxs.device)
        dx = torch.empty(batch * (seq or self.seq), self.width, device=dxs.device, dtype=torch.float32)
        check(lib.clipfs_tower_bwd_sparse(C.byref(t), dxs.data_ptr(), rows.data_ptr(), dx.data_ptr(), batch, saved.data_ptr(),
                                          scratch.data_ptr(), int(stop_at_input), torch.cuda.current_stream().cuda_stream),
              "tower_bwd_sparse")
        return dx


class Engine:
    def __init__(self, model):
        self.model = model
        v = model.visual
        self.vis = _TowerRT(v.transformer, v.tokens, stream0=1000)
        self.txt = _TowerRT(model.transformer, model.context_length, stream0=0)
        self.vproj_t = v.proj.data.t().contiguous()                  # [E, width]  (NT form of x @ proj)
        self.tproj_t = model.text_projection.data.t().contiguous()   # [E, width]
        self.seed_base = 0x5EED
        self.step = 0
        # Optional (off by default): run the text tower only on positions <= the last EOT of the batch.  Under the
        # causal mask nothing after a caption's EOT can influence its EOT feature, nor receive gradient from it, so
        # the reference's rows EOT+1..76 (jclip/model.py:202-215 encodes all 77) are dead work.  With LoRA dropout the
        # masks are indexed by token row = caption * (trimmed seq) + position: a trimmed run draws a different (equally
        # valid) mask stream than the untrimmed one, and LoRATrainer refuses trim_text + class-sharded text + dropout.
        self.trim_text = False
        self._trim_cache = {}
        # One row per sequence in the LAST block: the heads read the class token / the EOT token only, so after the last
        # attention everything is computed for that row alone -- forward (clipfs_tower_fwd_rows: output projection,
        # LayerNorm 2, MLP) and backward (clipfs_tower_bwd_sparse).  False = every block dense in both directions, the
        # gradient rows scattered into a zero-filled tensor (identical features and gradients; kept as the A/B
        # reference, bench.py reports it as a variant).  The name predates the forward half.
        self.sparse_backward = True

    @property
    def precision(self) -> str:
        return self.vis.precision

    @precision.setter
    def precision(self, mode: str) -> None:
        """"fp32" (default): every GEMM on the exact fp32 MFMA.  "bf16x3": the tower GEMMs (97 % of the FLOPs) run
        as split-bf16 with three bf16 MFMA products per operand pair, fp32 accumulate (weights are split once;
        logits move by ~2e-4 on 100 x cosine, inside the 1e-3 budget).  LayerNorm, attention, LoRA, the patch /
        projection / logits GEMMs and all reductions stay fp32.  "fp16": the same GEMMs with f16 operands
        (v_mfma_f32_32x32x16_f16, fp32 accumulate) -- cfg-5's "MFMA fp16 path"; tolerance ~1e-2 on the logits."""
        if mode not in ("fp32", "bf16x3", "fp16"):
            raise ValueError("precision must be 'fp32', 'bf16x3' or 'fp16'")
        self.vis.precision = mode
        self.txt.precision = mode

    # -- seeds ------------------------------------------------------------------------------------
    def next_seed(self) -> int:
        self.step += 1
        return _mix_seed(self.seed_base, self.step)

    # -- image tower --------------------------------------------------------------------------------
    def vit_forward(self, images: torch.Tensor, train: bool, seed: int = 0, own_saved: bool = False, row0: int = 0):
        """``row0``: index of images[0] in the GLOBAL batch (data-parallel shard) -- only the dropout masks see it."""
        m = self.model
        v = m.visual
        assert images.is_cuda and images.dtype == torch.float32, "images: fp32 device tensor [B,3,R,R]"
        images = images.contiguous()
        B = images.shape[0]
        if images.shape[1] != 3 or images.shape[2] != v.input_resolution or images.shape[3] != v.input_resolution:
            raise ValueError(f"expected images [B,3,{v.input_resolution},{v.input_resolution}], got {tuple(images.shape)}")
        L, d = v.tokens, v.width
        P = (v.input_resolution // v.patch_size) ** 2
        x0 = torch.empty(B * L, d, device=images.device, dtype=torch.float32)
        ops.patch_embed(images, v.conv1.weight.data, v.positional_embedding.data, x0, L)
        ops.vit_fill_special(x0, v.class_embedding.data, v.positional_embedding.data,
                             None if v.VPT is None else v.VPT.data, B, L, P)
        need_pre = train and v.VPT is not None
        if need_pre:
            x, mean0, rstd0 = ops.layernorm_fwd(x0, v.ln_pre.weight.data, v.ln_pre.bias.data, save_stats=True)
        else:
            x = ops.layernorm_fwd(x0, v.ln_pre.weight.data, v.ln_pre.bias.data)
            mean0 = rstd0 = None
        # only the class-token row of each image is read below (jclip/model.py:121-124)
        one_row = bool(self.sparse_backward)
        saved = self.vis.forward(x, B, train, seed, own_saved=own_saved, row0=row0,
                                 rows=self._class_rows(B, images.device) if one_row else None)
        if train:
            y, mean1, rstd1 = ops.layernorm_fwd(x, v.ln_post.weight.data, v.ln_post.bias.data, ldx=L * d, rows=B,
                                                save_stats=True)
        else:
            y = ops.layernorm_fwd(x, v.ln_post.weight.data, v.ln_post.bias.data, ldx=L * d, rows=B)
            mean1 = rstd1 = None
        feat = ops.gemm_nt(y, self.vproj_t)
        ctx = None
        if train:
            ctx = dict(B=B, x_final=x, saved=saved, stats=(mean1, rstd1), seed=seed, row0=row0, one_row=one_row,
                       pre=(x0, mean0, rstd0) if need_pre else None)
        return feat, ctx

    def vit_backward(self, ctx: dict, dfeat: torch.Tensor) -> None:
        """Accumulates into the LoRA gradient slots (and VPT.grad_slot)."""
        v = self.model.visual
        B, L, d = ctx["B"], v.tokens, v.width
        dy = ops.gemm_nt(dfeat.contiguous(), v.proj.data)  # [B, width] = dfeat @ proj^T
        mean1, rstd1 = ctx["stats"]
        # only the class-token row of each image carries gradient (jclip/model.py:121-124): it stays compact [B, width]
        # and the tower's last block works on B rows (clipfs_tower_bwd_sparse) -- no zero-filled [B*L, width] tensor
        dcls = ops.layernorm_bwd(dy, ctx["x_final"], v.ln_post.weight.data, mean1, rstd1, ldx=L * d)
        has_vpt = v.VPT is not None
        if ctx["one_row"]:  # the forward that produced ``saved`` decides (its last block kept one row per image)
            dx = self.vis.backward_sparse(dcls, self._class_rows(B, dfeat.device), B, ctx["saved"], ctx["seed"],
                                          stop_at_input=not has_vpt, row0=ctx["row0"])
        else:
            dx = ops.scatter_rows(dcls, self._class_rows(B, dfeat.device), L)
            self.vis.backward(dx, B, ctx["saved"], ctx["seed"], stop_at_input=not has_vpt, row0=ctx["row0"])
        if has_vpt:
            x0, mean0, rstd0 = ctx["pre"]
            dx0 = ops.layernorm_bwd(dx, x0, v.ln_pre.weight.data, mean0, rstd0)
            P = (v.input_resolution // v.patch_size) ** 2
            ops.token_rows_grad(dx0, v.VPT.grad_slot, B, L, 1 + P)

    def _class_rows(self, batch: int, device) -> torch.Tensor:
        z = getattr(self, "_zero_rows", None)
        if z is None or z.numel() < batch or z.device != device:
            z = torch.zeros(max(batch, 256), device=device, dtype=torch.int32)
            self._zero_rows = z
        return z[:batch]

    # -- text tower --------------------------------------------------------------------------------
    def _effective_ids(self, ids: torch.Tensor):
        """(ids possibly truncated to the batch's last EOT, effective sequence length)."""
        if not self.trim_text:
            return ids, ids.shape[1]
        key = (ids.data_ptr(), tuple(ids.shape), ids._version)
        hit = self._trim_cache.get(key)
        if hit is None:
            leff = int(ids.argmax(dim=-1).max().item()) + 1  # one host sync per distinct caption table
            hit = (ids[:, :leff].contiguous(), leff)
            self._trim_cache = {key: hit}
        return hit

    def text_forward(self, ids: torch.Tensor, prompt_ctx: Optional[torch.Tensor], train: bool, seed: int = 0,
                     own_saved: bool = False, row0: int = 0):
        m = self.model
        ids = ids.to(device=m.device, dtype=torch.int64).contiguous()
        n, seq = ids.shape
        if seq != m.context_length:
            raise ValueError(f"expected token ids [N,{m.context_length}], got {tuple(ids.shape)}")
        ids, seq = self._effective_ids(ids)
        x = ops.text_embed(ids, m.token_embedding.weight.data, m.positional_embedding.data,
                           None if prompt_ctx is None else prompt_ctx.data)
        # only the EOT row of each caption is read below (jclip/model.py:213-214)
        one_row = bool(self.sparse_backward)
        saved = self.txt.forward(x, n, train, seed, seq, own_saved=own_saved, row0=row0,
                                 rows=ops.eot_index(ids) if one_row else None)
        rows, idx = ops.gather_eot(x, ids)
        if train:
            y, mean, rstd = ops.layernorm_fwd(rows, m.ln_final.weight.data, m.ln_final.bias.data, save_stats=True)
        else:
            y = ops.layernorm_fwd(rows, m.ln_final.weight.data, m.ln_final.bias.data)
            mean = rstd = None
        feat = ops.gemm_nt(y, self.tproj_t)
        ctx = None
        if train:
            ctx = dict(n=n, seq=seq, rows=rows, idx=idx, stats=(mean, rstd), saved=saved, seed=seed, row0=row0,
                       one_row=one_row, has_ctx=prompt_ctx is not None)
        return feat, ctx

    def text_backward(self, ctx: dict, dfeat: torch.Tensor, dctx_slot: Optional[torch.Tensor] = None) -> None:
        m = self.model
        n, seq = ctx["n"], ctx["seq"]
        dy = ops.gemm_nt(dfeat.contiguous(), m.text_projection.data)
        mean, rstd = ctx["stats"]
        drows = ops.layernorm_bwd(dy, ctx["rows"], m.ln_final.weight.data, mean, rstd)
        # only the EOT row of each caption carries gradient (jclip/model.py:213-214)
        if ctx["one_row"]:
            dx = self.txt.backward_sparse(drows, ctx["idx"], n, ctx["saved"], ctx["seed"], stop_at_input=not ctx["has_ctx"],
                                          seq=seq, row0=ctx["row0"])
        else:
            dx = ops.scatter_rows(drows, ctx["idx"], seq)
            self.txt.backward(dx, n, ctx["saved"], ctx["seed"], stop_at_input=not ctx["has_ctx"], seq=seq, row0=ctx["row0"])
        if ctx["has_ctx"]:
            assert dctx_slot is not None
            ops.token_rows_grad(dx, dctx_slot, n, seq, 1)


# ------------------------------------------------------------------------------------------------
# autograd plumbing for the drop-in API (encode_image / encode_text return differentiable tensors)
# ------------------------------------------------------------------------------------------------

def _tower_trainables(tower_mod) -> List[Tuple[torch.nn.Parameter, torch.Tensor]]:
    """(parameter, gradient-slot view) pairs of the LoRA adapters of one tower."""
    out = []
    for blk in tower_mod.resblocks:
        a = blk.attn
        if getattr(a, "is_lora_mha", False):
            out.extend(a.trainable_pairs())
    return out


def _zero_slots(pairs):
    for _, g in pairs:
        g.zero_()


class _EncodeImage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, images, *params):
        eng = model.engine
        train = model.training
        seed = eng.next_seed() if train else 0
        feat, c = eng.vit_forward(images, True, seed, own_saved=True)
        ctx.model, ctx.c = model, c
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        model = ctx.model
        pairs = _tower_trainables(model.visual.transformer)
        _zero_slots(pairs)
        v = model.visual
        if v.VPT is not None:
            v.VPT.grad_slot.zero_()
        model.engine.vit_backward(ctx.c, dfeat.contiguous())
        grads = [g.clone() for _, g in pairs]
        if v.VPT is not None:
            grads.append(v.VPT.grad_slot.clone())
        return (None, None, *grads)


class _EncodeText(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, ids, prompt_ctx, *params):
        eng = model.engine
        seed = eng.next_seed() if model.training else 0
        feat, c = eng.text_forward(ids, prompt_ctx, True, seed, own_saved=True)
        ctx.model, ctx.c, ctx.prompt = model, c, prompt_ctx
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        model = ctx.model
        pairs = _tower_trainables(model.transformer)
        _zero_slots(pairs)
        slot = None
        if ctx.prompt is not None:
            slot = torch.zeros_like(ctx.prompt.data)
        model.engine.text_backward(ctx.c, dfeat.contiguous(), slot)
        return (None, None, slot, *[g.clone() for _, g in pairs])


def encode_image(model, images: torch.Tensor) -> torch.Tensor:
    images = images.to(device=model.device, dtype=torch.float32)
    pairs = _tower_trainables(model.visual.transformer)
    params = [p for p, _ in pairs]
    if model.visual.VPT is not None:
        _ensure_slot(model.visual.VPT)
        params.append(model.visual.VPT)
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return _EncodeImage.apply(model, images, *params)
    # no-grad route: dropout still follows model.training (reference: a second encode_image under jt.no_grad() in
    # train mode, slow_pace.py:1612,1659-1661, is dropout-on)
    feat, _ = model.engine.vit_forward(images, False, model.engine.next_seed() if model.training else 0)
    return feat


def encode_text(model, ids: torch.Tensor, prompt_ctx: Optional[torch.Tensor] = None) -> torch.Tensor:
    pairs = _tower_trainables(model.transformer)
    params = [p for p, _ in pairs]
    needs = any(p.requires_grad for p in params) or (prompt_ctx is not None and prompt_ctx.requires_grad)
    if torch.is_grad_enabled() and needs:
        return _EncodeText.apply(model, ids, prompt_ctx, *params)
    feat, _ = model.engine.text_forward(ids, prompt_ctx, False, model.engine.next_seed() if model.training else 0)
    return feat


def _ensure_slot(p: torch.nn.Parameter) -> None:
    if getattr(p, "grad_slot", None) is None or p.grad_slot.shape != p.shape:
        p.grad_slot = torch.zeros_like(p.data)


@torch.no_grad()
def clip_logits(model, image, text):
    """CLIP.execute (jclip/model.py:217-232): logit_scale.exp() * norm(img) @ norm(txt)^T."""
    fi = ops.l2norm_fwd(encode_image(model, image))
    ft = ops.l2norm_fwd(encode_text(model, text))
    li = ops.gemm_nt(fi, ft, alpha=float(model.logit_scale.data.exp().item()))
    return li, li.t()


# ------------------------------------------------------------------------------------------------
# differentiable head ops (each backed by a HIP kernel pair)
# ------------------------------------------------------------------------------------------------

class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, inv = ops.l2norm_fwd(x.contiguous(), save_inv=True)
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return ops.l2norm_bwd(dy.contiguous(), y, inv)


class _CosineLogits(torch.autograd.Function):
    """scale * a @ b^T, a [M,d], b [C,d]."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        ctx.scale = scale
        return ops.gemm_nt(a, b, alpha=scale)

    @staticmethod
    def backward(ctx, dl):
        a, b = ctx.saved_tensors
        dl = dl.contiguous()
        M, d = a.shape
        Cn = b.shape[0]
        da = ops.matmul_small(dl, b, M, d, Cn, Cn, 1, d, 1, ctx.scale)
        db = ops.matmul_small(dl, a, Cn, d, M, 1, Cn, d, 1, ctx.scale)
        return da, db, None


class _ClassMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, classes, templates):
        emb = emb.contiguous()
        ctx.save_for_backward(emb)
        ctx.dims = (classes, templates)
        return ops.class_mean_fwd(emb, classes, templates)

    @staticmethod
    def backward(ctx, dout):
        (emb,) = ctx.saved_tensors
        return ops.class_mean_bwd(emb, dout.contiguous(), *ctx.dims), None, None


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        loss_sum, dl, _ = ops.cross_entropy(logits.contiguous(), target, True, 1.0)
        ctx.save_for_backward(dl)
        return (loss_sum / logits.shape[0]).reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


def l2_normalize(x):
    return _L2Norm.apply(x)


def cosine_logits(a, b, scale=100.0):
    return _CosineLogits.apply(a, b, scale)


def class_mean(emb, classes, templates):
    return _ClassMean.apply(emb, classes, templates)


def cross_entropy_loss(logits, target):
    return _CrossEntropy.apply(logits, target.to(logits.device).long())
