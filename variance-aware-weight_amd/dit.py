"""DiT denoiser on hand-written gfx950 kernels, drop-in for the reference's `models/dit.py`.

Same constructor arguments, parameter names (state_dict keys), initialisation and call protocol as the
reference (`DiT.forward(x, t, y) -> (out, None)`, models/dit.py:157-280; presets :361-375), but nothing in
forward/backward is a torch op: the module owns a flat parameter buffer (flat.py) and drives libvaw_hip.so
through one autograd node whose backward is written out by hand (no autograd graph over the blocks).

Data layout in HBM (B images, T tokens, D hidden, M = B*T rows):
  residual stream   f32 [M, D]            one buffer per LayerNorm input (2 per block), kept for backward
  GEMM operands     act dtype [M, *]      bf16 in throughput mode, f32 in parity mode
  adaLN modulation  f32 [B, (6L+2) D]     ONE GEMM for all blocks + final layer; kernels take (ptr, row stride)
  weights           flat f32 master + flat bf16 shadow; all adaLN weights contiguous so that GEMM sees one matrix
  fp8 mode          (compute_dtype="fp8", BASELINE.json config 5) the four Linear layers of every block run on the scaled fp8
                    MFMA: activations / weights e4m3, gradients e5m2 (or e4m3), per-tensor scales taken on the device just before
                    use; every operand is kept k-major, so each tensor that feeds two contractions also has a transposed fp8
                    copy (W^T for dgrad; x^T, dy^T [*, M] per block for the deferred grouped weight gradients).  Everything else
                    -- attention, LayerNorm, embedders, final layer, adaLN, optimizer -- is the bf16 mode's.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from ._lib import BF16, F32, ptr
from .flat import FlatModule

__all__ = ["DiT", "DiT_S", "DiT_B", "DiT_L", "DiT_XL", "DiT_models"]


def _sincos_1d(dim, pos):
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    ang = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def get_2d_sincos_pos_embed(dim, grid_size):
    """Frozen 2-D sin-cos table, models/dit.py:307-354 (w index first, float32 grid, float64 angles)."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])
    return np.concatenate([_sincos_1d(dim // 2, grid[0]), _sincos_1d(dim // 2, grid[1])], axis=1)


class _Holder(nn.Module):
    """Parameter container that is never called."""

    def forward(self, *a, **k):
        raise RuntimeError("parameter holder: the DiT engine runs the arithmetic")


class _PatchEmbed(_Holder):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True)


class _TimestepEmbedder(_Holder):
    def __init__(self, hidden, freq=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(freq, hidden), nn.SiLU(), nn.Linear(hidden, hidden))
        self.frequency_embedding_size = freq


class _LabelEmbedder(_Holder):
    def __init__(self, num_classes, hidden, dropout_prob):
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes + int(dropout_prob > 0), hidden)
        self.num_classes, self.dropout_prob = num_classes, dropout_prob


class _Attention(_Holder):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(_Holder):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU(approximate="tanh")
        self.fc2 = nn.Linear(hidden, dim)


class _Block(_Holder):
    def __init__(self, dim, heads, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(dim, 6 * dim, bias=True))


class _FinalLayer(_Holder):
    def __init__(self, dim, patch, out_ch):
        super().__init__()
        self.norm_final = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(dim, patch * patch * out_ch, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(dim, 2 * dim, bias=True))


class _Workspace:
    """Activation buffers for one batch size; reused every step (no allocator traffic in the loop)."""

    def __init__(self, m, B, adt):
        dev = m._flat.device
        f32, T, D, Lyr = torch.float32, m.T, m.D, m.depth
        M = B * T
        e = lambda *s, dtype=adt: torch.empty(*s, device=dev, dtype=dtype)
        self.B, self.adt, self.gen = B, adt, 0
        # The conditioning path's weight gradients are dy^T x products with K = BATCH.  The MFMA GEMMs take K in whole 64s, so the
        # bf16 operands of those products (tfreq, h1s, cs here; dmod_a, dc_a, dh1_a below) carry zero rows up to the next multiple of
        # 64: nothing ever writes them, the products run with K = Bk on the MFMA kernels instead of the generic one (32 images per
        # GPU: 0.19 ms of the 4.5 ms step), and the bias gradients (column sums) still walk the first B rows only.
        self.Bk = Bk = ((B + 63) // 64) * 64 if adt == torch.bfloat16 else B
        z = lambda cols: torch.zeros(Bk, cols, device=dev, dtype=adt)
        self.tfreq, self.h1, self.h1s = z(256), e(B, D, dtype=f32), z(D)
        self.temb, self.c, self.cs = e(B, D, dtype=f32), e(B, D, dtype=f32), z(D)
        self.mod = e(B, m.mod_cols, dtype=f32)
        self.xp = e(M, m.Kp)
        self.xres = [e(M, D, dtype=f32) for _ in range(2 * Lyr + 1)]   # inputs of LN1/LN2 of each block, + final
        self.blk = [dict(xm=e(M, D), qkv=e(M, 3 * D), ao=e(M, D), lse=e(B * m.num_heads * T, dtype=f32), y1=e(M, D),
                         xm2=e(M, D), hpre=e(M, m.Dm), a=e(M, m.Dm), y2=e(M, D),
                         mean1=e(M, dtype=f32), rstd1=e(M, dtype=f32), mean2=e(M, dtype=f32), rstd2=e(M, dtype=f32))
                    for _ in range(Lyr)]
        self.xf, self.meanf, self.rstdf = e(M, D), e(M, dtype=f32), e(M, dtype=f32)
        self.otok = e(M, m.No, dtype=f32)
        # backward scratch (shared by all blocks)
        self.dotok, self.dres, self.dD = e(M, m.No), e(M, D, dtype=f32), e(M, D)
        self.dDm, self.dqkv, self.dao = e(M, m.Dm), e(M, 3 * D), e(M, D)
        # deferred weight gradients (bf16 mode): every block keeps the four dy operands of its Linear layers until ONE grouped
        # launch at the end of backward consumes them (226 MB per DiT-B/4 block at batch 256; sized for 288 GB of HBM)
        self.fp8 = bool(m._fp8)
        self.defer = (bool(m.defer_wgrad) or self.fp8) and adt == torch.bfloat16 and M % 64 == 0
        if self.fp8:
            # per block: the transposed fp8 copies the grouped weight gradients read (x^T and dy^T, [features][M]); the row-major
            # copies live in a shared scratch just long enough for the GEMM that follows the quantiser
            gf = L.BF8 if m.fp8_grad_format == "e5m2" else L.FP8
            fmts = [L.FP8] * 4 + [gf] * 4
            self.fp8_states = ops.fp8_states(fmts * Lyr, dev, m.fp8_margin)     # delayed scaling: one update launch per step
            self.calib_fwd = self.calib_bwd = False                             # first pass: scales taken just in time
            self.d_fwd = self.d_bwd = False
            for l, b in enumerate(self.blk):
                F = lambda i, cols: ops.Fp8(M, cols, dev, transposed=True, plain=False, fmt=fmts[i], state=self.fp8_states[8 * l + i])
                b.update(f_xm=F(0, D), f_ao=F(1, D), f_xm2=F(2, D), f_a=F(3, m.Dm), f_dy2=F(4, D), f_dhid=F(5, m.Dm), f_dy1=F(6, D),
                         f_dqkv=F(7, 3 * D))
        elif self.defer:
            for b in self.blk:
                b.update(dy2=e(M, D), dDm=e(M, m.Dm), dy1=e(M, D), dqkv=e(M, 3 * D))
        self.wgrad_groups = {}
        # bias gradients: the kernels that produce dy leave partial column sums per block (gate backward: one row per sample;
        # fc2's GELU' epilogue: one per 64 or 128 token rows; attention backward: one per sample and 64-token block); ONE batched fold
        # per group of blocks turns them into the four bias gradients of every block (ops.ReduceGroup)
        self.bias_groups = {}
        for b in self.blk:
            b.update(cp_fc2=ops.ColsumPartial(B, D, dev), cp_proj=ops.ColsumPartial(B, D, dev),
                     cp_fc1=ops.ColsumPartial((M + 63) // 64, m.Dm, dev), cp_qkv=ops.ColsumPartial(max(B, M // 64), 3 * D, dev))
            b["cp_fc2"].rows.value = b["cp_proj"].rows.value = B
        self.dyb = e(M, D)          # dy of a gated branch when the block keeps no buffer of its own (f32 / fp8 / per-layer wgrad modes)
        self.delta = e(B * m.num_heads * T, dtype=f32)
        self.dmod, self.dmod_a = e(B, m.mod_cols, dtype=f32), z(m.mod_cols)
        self.dcs, self.dc, self.dc_a = e(B, D, dtype=f32), e(B, D, dtype=f32), z(D)
        self.dh1s, self.dh1, self.dh1_a = e(B, D, dtype=f32), e(B, D, dtype=f32), z(D)
        self.dxp = e(M, m.Kp, dtype=f32)


class _DiTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, x, t, y):
        out = model._forward_impl(x, t, y)
        ctx.model, ctx.gen, ctx.need_dx = model, model._ws_cur.gen, x.requires_grad
        ctx.x_shape = x.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.model
        if m._ws_cur.gen != ctx.gen:
            raise L.VawError("DiT backward: activations were overwritten by a later forward of the same batch size")
        dx = m._backward_impl(dout.contiguous(), ctx.need_dx)
        return torch.zeros_like(m._anchor), None, dx, None, None


class DiT(FlatModule):
    def __init__(self, image_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, class_dropout_prob=0.1, num_classes=1000, learn_sigma=False, learn_align=False,
                 encoder_depth=8, z_dims=768, projector_dim=2048, compute_dtype="bf16"):
        super().__init__()
        if learn_align:
            raise NotImplementedError("learn_align (REPA feature alignment) is outside the hot path (SURVEY §2.1 row 12)")
        assert hidden_size % num_heads == 0 and hidden_size % 4 == 0
        self.learn_sigma, self.learn_align = learn_sigma, learn_align
        self.in_channels = in_channels
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.patch_size, self.num_heads, self.depth = patch_size, num_heads, depth
        self.image_size = image_size
        self.encoder_depth = encoder_depth
        # same registration order as the reference => same weights under the same torch seed
        self.x_embedder = _PatchEmbed(image_size, patch_size, in_channels, hidden_size)
        self.t_embedder = _TimestepEmbedder(hidden_size)
        self.y_embedder = _LabelEmbedder(num_classes, hidden_size, class_dropout_prob)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.x_embedder.num_patches, hidden_size), requires_grad=False)
        self.blocks = nn.ModuleList([_Block(hidden_size, num_heads, mlp_ratio) for _ in range(depth)])
        self.projectors = None
        self.final_layer = _FinalLayer(hidden_size, patch_size, self.out_channels)
        self.initialize_weights()
        # engine geometry
        self.D, self.T = hidden_size, self.x_embedder.num_patches
        self.Dm = int(hidden_size * mlp_ratio)
        self.Kp = in_channels * patch_size * patch_size
        self.No = patch_size * patch_size * self.out_channels
        self.mod_cols = (6 * depth + 2) * hidden_size
        self.fp8_grad_format = "e5m2"      # fp8 mode: gradients in e5m2 (range) or "e4m3" (precision); weights / activations e4m3
        # fp8 scales: "delayed" = each tensor's scale comes from its max |x| of the previous step (x margin), taken by the
        # quantiser itself -- one pass per tensor; the first step, and "jit" always, measure the tensor first (two passes)
        self.fp8_scaling, self.fp8_margin = "delayed", 2.0
        # with delayed scaling, fc1's GELU epilogue and fc2's GELU' epilogue write their results (`a`, d hidden) as fp8 themselves --
        # byte-identical to quantising the bf16 tensor, which is then never written or read
        self.fp8_fuse_epilogue = os.environ.get("VAW_FP8_FUSE", "1") != "0"
        self.fp8_fuse_rows = os.environ.get("VAW_FP8_FUSE_ROWS", "1") != "0"      # ... and LN-modulate / gate backward their outputs
        self._fp8_w, self._fp8_epoch, self._fp8_wstates, self._fp8_wgroup = {}, None, None, None
        self.set_compute_dtype(compute_dtype)
        self._anchor = torch.zeros(1, requires_grad=True)
        self._ws, self._ws_cur = {}, None
        self.grad_ready_hook = None      # callable(stage:int) -> None; stage counts down from depth+1 to 0
        # bf16 mode: the weight gradients of the blocks' Linear layers are deferred and run as grouped launches
        # (vaw_wgrad_grouped): one for all blocks at the end of backward, or two (upper / lower half of the blocks) when a
        # gradient hook listens, so that the first half's all-reduce still overlaps the rest of backward.  False: per layer.
        self.defer_wgrad = True

    # ---- reference surface -----------------------------------------------------------------
    def initialize_weights(self):
        """models/dit.py:206-241 (xavier on every Linear, adaLN-Zero, frozen sin-cos pos_embed)."""
        def basic(mod):
            if isinstance(mod, nn.Linear):
                nn.init.xavier_uniform_(mod.weight)
                if mod.bias is not None:
                    nn.init.constant_(mod.bias, 0)
        self.apply(basic)
        pe = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.x_embedder.num_patches ** 0.5))
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        w = self.x_embedder.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.constant_(self.x_embedder.proj.bias, 0)
        nn.init.normal_(self.y_embedder.embedding_table.weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        for blk in self.blocks:
            nn.init.constant_(blk.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(blk.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].weight, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)

    def set_compute_dtype(self, name):
        """'bf16' (throughput: bf16 MFMA, f32 accumulate), 'fp32' (parity with the CPU reference) or 'fp8' (the blocks' Linear
        layers on the scaled fp8 MFMA with f32 accumulate, everything else as in 'bf16')."""
        name = {"float32": "fp32", "f32": "fp32", "bfloat16": "bf16", "float8": "fp8"}.get(name, name)
        if name not in ("bf16", "fp32", "fp8"):
            raise ValueError(f"compute_dtype must be 'bf16', 'fp32' or 'fp8', got {name}")
        if name == "fp8" and (self.D % 128 or self.Dm % 128):
            raise ValueError("compute_dtype='fp8' needs hidden and MLP widths that are multiples of 128 (one fp8 MFMA K tile)")
        if name != getattr(self, "compute_dtype", name):
            self.require_fresh_masters("set_compute_dtype()")      # (every other mode derives its weights from the f32 masters)
        self.compute_dtype = name
        self._dt = F32 if name == "fp32" else BF16
        self._fp8 = name == "fp8"
        self._fp8_w, self._fp8_epoch, self._fp8_wstates, self._fp8_wgroup = {}, None, None, None
        self._ws = {}

    def _flat_groups(self):
        ada_w = [f"blocks.{i}.adaLN_modulation.1.weight" for i in range(self.depth)] + ["final_layer.adaLN_modulation.1.weight"]
        ada_b = [f"blocks.{i}.adaLN_modulation.1.bias" for i in range(self.depth)] + ["final_layer.adaLN_modulation.1.bias"]
        return [ada_w, ada_b]

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        self._anchor = fn(self._anchor.detach()).requires_grad_(True)
        self._ws = {}
        return r

    def grad_stage_bounds(self):
        """stage -> [start, end) of the flat gradient buffer that is final when backward reports `stage`
        (depth+1: head done; l+1: block l done; 0: embedders and adaLN done).  Used for all-reduce buckets."""
        self.ensure_flat()
        off = lambda n: self._flat_offsets[n][0]
        first = [off(f"blocks.{l}.attn.qkv.weight") for l in range(self.depth)] + [off("final_layer.linear.weight")]
        b = {self.depth + 1: (first[-1], self._flat_n_train), 0: (0, first[0])}
        for l in range(self.depth):
            b[l + 1] = (first[l], first[l + 1])
        # the packed adaLN matrix is the largest tensor of the embedder stage (DiT-B: 42 M of its 44 M parameters) and
        # would be reduced after everything else, un-overlapped.  Its rows for blocks >= depth/2 (and the final layer's)
        # are final half-way through backward: stage "ada_hi" ships them then; stage 0 keeps only the lower rows.
        half = self._ada_split_block()
        if half is not None:
            D = self.D
            wo, bo = off("blocks.0.adaLN_modulation.1.weight"), off("blocks.0.adaLN_modulation.1.bias")
            we = sum(self._flat_offsets["final_layer.adaLN_modulation.1.weight"])       # end of the packed [(6L+2)D, D] matrix
            be = sum(self._flat_offsets["final_layer.adaLN_modulation.1.bias"])
            assert we - wo == (6 * self.depth + 2) * D * D and be - bo == (6 * self.depth + 2) * D
            cut_w, cut_b = wo + 6 * half * D * D, bo + 6 * half * D
            b["ada_hi"] = [(cut_w, we), (cut_b, be)]
            lo = [(0, cut_w), (we, cut_b), (be, first[0])] if wo < bo else [(0, cut_b), (be, cut_w), (we, first[0])]
            b[0] = [r for r in lo if r[1] > r[0]]
        return b

    def _ada_split_block(self):
        """First block whose adaLN rows go out with the early bucket (None: no split -- shallow models)."""
        return self.depth // 2 if self.depth >= 4 else None

    def unpatchify(self, x):
        c, p = self.out_channels, self.patch_size
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.float().contiguous()
        img = torch.empty(x.shape[0], c, h * p, w * p, device=x.device, dtype=torch.float32)
        L.check(L.lib().vaw_unpatchify(F32, ptr(x), ptr(img), x.shape[0], c, h * p, w * p, p, L.stream_ptr()), "vaw_unpatchify")
        return img

    def forward(self, x, t, y, **kwargs):
        """x [N,C,H,W] f32, t [N] (float timesteps, already rescaled), y [N] int64 -> (eps [N,C_out,H,W] f32, None)."""
        L.need_cuda(x, t, y)
        out = _DiTFn.apply(self._anchor, self, x, t, y)
        return out, None

    # ---- engine ----------------------------------------------------------------------------
    stagewise_weight_waits = True      # parallel.DistributedDataParallel (sharded optimizer): forward asks for its weights per stage

    def _need(self, stage):
        """The gathered weights of gradient stage `stage` (grad_stage_bounds) are about to be read: with a sharded optimizer the
        all-gather that brings them may still be in flight on the collective stream."""
        z = getattr(self, "_zero", None)
        if z is not None:
            z.wait_stage(stage)

    def _w(self, name):
        """device address of a parameter in the dtype the kernels read (bf16 shadow or f32 master)."""
        o, _ = self._flat_offsets[name]
        return self._wbase + o * self._wsize

    def _g(self, name):
        o, _ = self._flat_offsets[name]
        return self._gbase + 4 * o

    def _p32(self, name):
        o, _ = self._flat_offsets[name]
        return self._flat.data_ptr() + 4 * o

    def _prepare(self, B):
        self.ensure_flat()
        if self._dt == BF16:
            sh = self.shadow_bf16()
            self._wbase, self._wsize = sh.data_ptr(), 2
        else:
            self._wbase, self._wsize = self._flat.data_ptr(), 4
        adt = L.TORCH_DTYPE[self._dt]
        if self._fp8:
            if (B * self.T) % 128:
                raise L.VawError("compute_dtype='fp8': batch * tokens must be a multiple of 128 (K tile of the weight gradients)")
            self._refresh_fp8_weights()
        key = (B, adt, self._fp8)
        if key not in self._ws:
            self._ws[key] = _Workspace(self, B, adt)
        self._ws_cur = ws = self._ws[key]
        if ws.fp8:
            delayed = self.fp8_scaling == "delayed"
            ws.d_fwd, ws.d_bwd = delayed and ws.calib_fwd, delayed and ws.calib_bwd
            if ws.d_fwd or ws.d_bwd:
                ops.fp8_scale_update(ws.fp8_states)
        return ws

    _FP8_LINEARS = (("attn.qkv.", 3, 1), ("attn.proj.", 1, 1), ("mlp.fc1.", 0, 1), ("mlp.fc2.", 1, 0))   # (name, N / D or 0 = Dm, K / D or 0 = Dm)

    def _refresh_fp8_weights(self):
        """e4m3 copies W [N][K] and W^T [K][N] of the blocks' Linear weights, re-quantised from the f32 masters whenever the
        weights changed (optimizer step, load_state_dict, broadcast)."""
        epoch = (self._weights_epoch, self._flat.data_ptr())
        if self._fp8_epoch == epoch:
            return
        self.require_fresh_masters("re-quantising the fp8 weights")
        dev = self._flat.device
        if self._fp8_wstates is None or self._fp8_wstates.device != dev:
            self._fp8_wstates, self._fp8_w, self._fp8_wgroup = ops.fp8_states([L.FP8] * (4 * self.depth), dev, self.fp8_margin), {}, None
        delayed = self.fp8_scaling == "delayed" and bool(self._fp8_w)
        if delayed:
            ops.fp8_scale_update(self._fp8_wstates)
            grp = getattr(self, "_fp8_wgroup", None)
            if grp is None or grp[0] != self._flat.data_ptr():
                pairs = [(self._p32(f"blocks.{l}.{nm}weight"), self._fp8_w[f"blocks.{l}.{nm}weight"])
                         for l in range(self.depth) for nm, _, _ in self._FP8_LINEARS]
                grp = self._fp8_wgroup = (self._flat.data_ptr(), ops.Fp8QuantGroup(pairs, dev))
            grp[1].launch()          # all 4 L weight matrices in one launch
            self._fp8_epoch = epoch
            return
        i = 0
        for l in range(self.depth):
            for nm, nf, kf in self._FP8_LINEARS:
                name = f"blocks.{l}.{nm}weight"
                N, K = (nf * self.D or self.Dm), (kf * self.D or self.Dm)
                f = self._fp8_w.get(name)
                if f is None:
                    f = self._fp8_w[name] = ops.Fp8(N, K, dev, state=self._fp8_wstates[i])
                f.quantize(self._p32(name), src_dt=F32, delayed=False)
                i += 1
        self._fp8_epoch = epoch

    def _forward_impl(self, x, t, y):
        B, C, H, W = x.shape
        assert C == self.in_channels and H == W == self.image_size, f"expected [N,{self.in_channels},{self.image_size},{self.image_size}], got {tuple(x.shape)}"
        assert t.shape == (B,) and y.shape == (B,) and y.dtype == torch.int64
        ws = self._prepare(B)
        ws.gen += 1
        dt, D, T, Dm, Lyr, ld = self._dt, self.D, self.T, self.Dm, self.depth, self.mod_cols
        M, lib, st = B * T, L.lib(), L.stream_ptr()
        x = x.float().contiguous()
        tf = t.float().contiguous()
        ye = self.y_embedder
        if (self.training and ye.dropout_prob > 0):
            drop = torch.rand(B, device=y.device) < ye.dropout_prob      # reference: dit.py:94-103
            y = torch.where(drop, ye.num_classes, y)
        ws.y = y.contiguous()
        # conditioning vector c = t_emb + y_emb, then every block's modulation in one GEMM
        self._need(0)
        L.check(lib.vaw_timestep_embedding(dt, ptr(tf), ptr(ws.tfreq), B, 256, 10000.0, st), "timestep_embedding")
        ops.gemm(dt, 1, 1, B, D, 256, ptr(ws.tfreq), 256, self._w("t_embedder.mlp.0.weight"), 256, ptr(ws.h1), D,
                 bias=self._p32("t_embedder.mlp.0.bias"), out_f32=True)
        L.check(lib.vaw_silu_fwd(dt, ptr(ws.h1), ptr(ws.h1s), B * D, st), "silu")
        ops.gemm(dt, 1, 1, B, D, D, ptr(ws.h1s), D, self._w("t_embedder.mlp.2.weight"), D, ptr(ws.temb), D,
                 bias=self._p32("t_embedder.mlp.2.bias"), out_f32=True)
        L.check(lib.vaw_add_embedding(ptr(ws.temb), self._p32("y_embedder.embedding_table.weight"), ptr(ws.y), ptr(ws.c),
                                      B, D, ye.embedding_table.num_embeddings, st), "add_embedding")
        L.check(lib.vaw_silu_fwd(dt, ptr(ws.c), ptr(ws.cs), B * D, st), "silu")
        self._need("ada_hi")
        ops.gemm(dt, 1, 1, B, ld, D, ptr(ws.cs), D, self._w("blocks.0.adaLN_modulation.1.weight"), D, ptr(ws.mod), ld,
                 bias=self._p32("blocks.0.adaLN_modulation.1.bias"), out_f32=True)
        # tokens
        L.check(lib.vaw_patchify(dt, ptr(x), ptr(ws.xp), B, C, H, W, self.patch_size, st), "patchify")
        ops.gemm(dt, 1, 1, M, D, self.Kp, ptr(ws.xp), self.Kp, self._w("x_embedder.proj.weight"), self.Kp, ptr(ws.xres[0]), D,
                 bias=self._p32("x_embedder.proj.bias"), rowadd=self._p32("pos_embed"), rows_per_batch=T, out_f32=True)
        mod = ptr(ws.mod)
        for l in range(Lyr):
            b, pre = ws.blk[l], f"blocks.{l}."
            mo = mod + 4 * (6 * l * D)
            xin, xmid, xout = ptr(ws.xres[2 * l]), ptr(ws.xres[2 * l + 1]), ptr(ws.xres[2 * l + 2])
            self._need(l + 1)
            fuse_rows = ws.fp8 and ws.d_fwd and self.fp8_fuse_epilogue and self.fp8_fuse_rows and M % 64 == 0 and D % 128 == 0    # LN writes fp8 itself
            if fuse_rows:
                ops.ln_modulate_fwd_fp8(xin, mo, mo + 4 * D, ld, b["f_xm"], ptr(b["mean1"]), ptr(b["rstd1"]), B, T, D)
            else:
                ops.ln_modulate_fwd(dt, xin, mo, mo + 4 * D, ld, ptr(b["xm"]), ptr(b["mean1"]), ptr(b["rstd1"]), B, T, D)
            self._linear_fwd(ws, b, "f_xm", None if fuse_rows else b["xm"], pre + "attn.qkv.", M, 3 * D, D, ptr(b["qkv"]), 3 * D)
            q = ptr(b["qkv"])
            es = self._wsize
            ops.attn_fwd(dt, self._attn_desc(B), q, q + es * D, q + 2 * es * D, ptr(b["ao"]), ptr(b["lse"]))
            self._linear_fwd(ws, b, "f_ao", b["ao"], pre + "attn.proj.", M, D, D, xmid, D, aux_out=ptr(b["y1"]), gate=mo + 4 * 2 * D,
                             gate_ld=ld, resid=xin, rows_per_batch=T, out_f32=True)
            if fuse_rows:
                ops.ln_modulate_fwd_fp8(xmid, mo + 4 * 3 * D, mo + 4 * 4 * D, ld, b["f_xm2"], ptr(b["mean2"]), ptr(b["rstd2"]), B, T, D)
            else:
                ops.ln_modulate_fwd(dt, xmid, mo + 4 * 3 * D, mo + 4 * 4 * D, ld, ptr(b["xm2"]), ptr(b["mean2"]), ptr(b["rstd2"]), B, T, D)
            fuse_a = ws.fp8 and ws.d_fwd and self.fp8_fuse_epilogue and M % 64 == 0      # fc1's epilogue writes `a` as e4m3 itself
            self._linear_fwd(ws, b, "f_xm2", None if fuse_rows else b["xm2"], pre + "mlp.fc1.", M, Dm, D, b["f_a"].epilogue_target() if fuse_a else ptr(b["a"]), Dm,
                             act=1, aux_out=ptr(b["hpre"]), **({"out_fp8": b["f_a"]} if fuse_a else {}))
            self._linear_fwd(ws, b, "f_a", None if fuse_a else b["a"], pre + "mlp.fc2.", M, D, Dm, xout, D, aux_out=ptr(b["y2"]),
                             gate=mo + 4 * 5 * D, gate_ld=ld, resid=xmid, rows_per_batch=T, out_f32=True)
        mo = mod + 4 * (6 * Lyr * D)
        self._need(Lyr + 1)
        ops.ln_modulate_fwd(dt, ptr(ws.xres[2 * Lyr]), mo, mo + 4 * D, ld, ptr(ws.xf), ptr(ws.meanf), ptr(ws.rstdf), B, T, D)
        ops.gemm(dt, 1, 1, M, self.No, D, ptr(ws.xf), D, self._w("final_layer.linear.weight"), D, ptr(ws.otok), self.No,
                 bias=self._p32("final_layer.linear.bias"), out_f32=True)
        out = torch.empty(B, self.out_channels, H, W, device=x.device, dtype=torch.float32)
        L.check(lib.vaw_unpatchify(dt, ptr(ws.otok), ptr(out), B, self.out_channels, H, W, self.patch_size, st), "unpatchify")
        if ws.fp8:
            ws.calib_fwd = True
        return out

    def _linear_fwd(self, ws, b, fkey, x, name, M, N, K, out, ldc, **epi):
        """y = x W^T + bias with the block's epilogue: bf16 / f32 MFMA GEMM, or (fp8 mode) quantise x (keeping x^T for the weight
        gradient) and run the e4m3 x e4m3 GEMM."""
        if not ws.fp8:
            ops.gemm(self._dt, 1, 1, M, N, K, ptr(x), K, self._w(name + "weight"), K, out, ldc, bias=self._p32(name + "bias"), **epi)
            return
        # x is None: the producing GEMM's epilogue already left the row-major fp8 bytes (and the running max) in b[fkey]
        f = b[fkey].quantize(x, delayed=ws.d_fwd) if x is not None else b[fkey].transpose_from_q()
        w = self._fp8_w[name + "weight"]
        ops.gemm_fp8(M, N, K, f.last_q, K, ptr(f.scale), ptr(w.q), K, ptr(w.scale), out, ldc, bias=self._p32(name + "bias"), **epi)

    def _linear_dgrad(self, ws, b, fkey, dy, name, M, N, K, out, **epi):
        """dx[M,K] = dy[M,N] W[N,K] (+ epilogue).  fp8 mode: quantise dy (keeping dy^T for the weight gradient), contract with W^T."""
        if not ws.fp8:
            ops.gemm(self._dt, 1, 0, M, K, N, ptr(dy) if isinstance(dy, torch.Tensor) else dy, N, self._w(name + "weight"), K, out, K, **epi)
            return
        f = b[fkey].quantize(dy, src_dt=BF16, delayed=ws.d_bwd) if dy is not None else b[fkey].transpose_from_q()
        w = self._fp8_w[name + "weight"]
        ops.gemm_fp8(M, K, N, f.last_q, N, ptr(f.scale), ptr(w.qt), N, ptr(w.scale), out, K, a_format=f.fmt, **epi)

    def _attn_desc(self, B):
        d = getattr(self, "_adesc", None)
        if d is None or d.B != B:
            d = self._adesc = ops.attn_desc_token_major(B, self.num_heads, self.T, self.D // self.num_heads)
        return d

    def _wgrad(self, dt, name, dy, x, Nw, Kw, M, beta, bias=True):
        """dW[Nw,Kw] (+)= dy[M,Nw]^T x[M,Kw];  db[Nw] (+)= colsum(dy), on the same launch (row sums of the staged dy^T
        tiles), unless the producer of dy already delivered it (bias=False).  name = '<module>.' prefix."""
        ops.gemm(dt, 0, 0, Nw, Kw, M, dy, Nw, x, Kw, self._g(name + "weight"), Kw, beta=beta, out_f32=True,
                 rowsum_a_out=self._g(name + "bias") if bias else None, rowsum_a_beta=beta)

    def _adaln_wgrad(self, dt, ws, r0, nrows, B, beta):
        """Weight and bias gradient of rows [r0, r0 + nrows) of the packed adaLN matrix: dW = dmod[:, rows]^T cs, db = colsum.
        Returns the address of dmod in the GEMM operand dtype (whole buffer)."""
        D, ld = self.D, self.mod_cols
        if dt == BF16:
            ops.cast_bf16(ws.dmod, ws.dmod_a)
            dmod_a, es = ptr(ws.dmod_a), 2
        else:
            dmod_a, es = ptr(ws.dmod), 4
        ops.gemm(dt, 0, 0, nrows, D, ws.Bk, dmod_a + es * r0, ld, ptr(ws.cs), D, self._g("blocks.0.adaLN_modulation.1.weight") + 4 * r0 * D,
                 D, beta=beta, out_f32=True)                 # K = the batch, zero rows up to ws.Bk (see _Workspace)
        ops.colsum(dt, dmod_a + es * r0, B, nrows, ld, self._g("blocks.0.adaLN_modulation.1.bias") + 4 * r0, beta)
        return dmod_a

    def _backward_impl(self, dout, need_dx):
        ws = self._ws_cur
        B = ws.B
        dt, D, T, Dm, Lyr, ld = self._dt, self.D, self.T, self.Dm, self.depth, self.mod_cols
        M, lib, st = B * T, L.lib(), L.stream_ptr()
        H = W = self.image_size
        beta = 1.0 if self.grads_live() else 0.0
        self._gbase = self.flat_grads().data_ptr()
        hook = self.grad_ready_hook
        ada_half = self._ada_split_block() if hook else None      # early adaLN bucket only when somebody listens (DDP)
        dres, dD, dDm, dmod = ptr(ws.dres), ptr(ws.dD), ptr(ws.dDm), ptr(ws.dmod)
        mod = ptr(ws.mod)
        es = self._wsize
        defer = ws.defer and dt == BF16
        fp8 = ws.fp8
        own_dy = defer and not fp8            # every block keeps its four dy operands for the grouped weight-gradient launch
        # fp8, delayed scaling: the row kernels write dy as fp8 themselves
        fuse_rows = fp8 and ws.d_bwd and self.fp8_fuse_epilogue and self.fp8_fuse_rows and M % 64 == 0 and D % 128 == 0

        def ln_bwd_and_gate(dout_p, x_p, mean_p, rstd_p, scale_p, dres_in, dsh, dsc, nxt):
            """Backward of one LayerNorm+modulate (d of its output -> residual-stream gradient, dshift, dscale) and, in the same
            pass over the rows, the gate backward of the branch in FRONT of it (nxt = (block index, "mlp" | "attn") or None):
            dy of that branch, its dgate and the per-sample column sums of dy (the bias gradient of fc2 / proj)."""
            if nxt is None:
                ops.ln_modulate_bwd(dt, dout_p, x_p, mean_p, rstd_p, scale_p, ld, dres_in, dres, dsh, dsc, ld, B, T, D)
                return
            l2, which = nxt
            nb = ws.blk[l2]
            mo2, dmo2 = mod + 4 * (6 * l2 * D), dmod + 4 * (6 * l2 * D)
            gcol = 5 if which == "mlp" else 2
            y, cp = (nb["y2"], nb["cp_fc2"]) if which == "mlp" else (nb["y1"], nb["cp_proj"])
            if fuse_rows:
                ops.ln_modulate_bwd_gate_fp8(dout_p, x_p, mean_p, rstd_p, scale_p, ld, dres_in, dres, dsh, dsc, ld, ptr(y),
                                             mo2 + 4 * gcol * D, nb["f_dy2" if which == "mlp" else "f_dy1"], dmo2 + 4 * gcol * D,
                                             B, T, D, cp.buf.data_ptr())
            else:
                dy = ptr(nb["dy2" if which == "mlp" else "dy1"]) if own_dy else ptr(ws.dyb)
                ops.ln_modulate_bwd_gate(dt, dout_p, x_p, mean_p, rstd_p, scale_p, ld, dres_in, dres, dsh, dsc, ld, ptr(y),
                                         mo2 + 4 * gcol * D, dy, dmo2 + 4 * gcol * D, B, T, D, cp.buf.data_ptr())

        # head: unpatchify^T, final linear, final LN+modulate (+ the gate backward of the last block's MLP branch)
        L.check(lib.vaw_unpatchify_bwd(dt, ptr(dout), ptr(ws.dotok), B, self.out_channels, H, W, self.patch_size, st), "unpatchify_bwd")
        self._wgrad(dt, "final_layer.linear.", ptr(ws.dotok), ptr(ws.xf), self.No, D, M, beta)
        ops.gemm(dt, 1, 0, M, D, self.No, ptr(ws.dotok), self.No, self._w("final_layer.linear.weight"), D, dD, D)
        mo, dmo = mod + 4 * (6 * Lyr * D), dmod + 4 * (6 * Lyr * D)
        ln_bwd_and_gate(dD, ptr(ws.xres[2 * Lyr]), ptr(ws.meanf), ptr(ws.rstdf), mo + 4 * D, 0, dmo, dmo + 4 * D,
                        (Lyr - 1, "mlp") if Lyr else None)
        if hook:
            hook(Lyr + 1)
        pending = []                      # blocks whose weight gradients wait for the next grouped launch
        group_cut = Lyr // 2 if (hook and Lyr >= 4) else 0      # with a listener: flush once half-way, once at the end

        def fold_bias(blocks, qkv_too):
            """The bias gradients of `blocks` from the partial column sums their backward left behind: one launch."""
            key = (blocks[0], blocks[-1], self._gbase, qkv_too)
            cps = []
            for l in blocks:
                b, pre = ws.blk[l], f"blocks.{l}."
                names = [("cp_fc2", "mlp.fc2."), ("cp_fc1", "mlp.fc1."), ("cp_proj", "attn.proj.")] + ([("cp_qkv", "attn.qkv.")] if qkv_too else [])
                cps += [(b[key_cp], pre + nm) for key_cp, nm in names]
            # the number of partial rows a producer leaves is NOT a function of the shape alone: vaw_gemm writes one row per 64, 128
            # or 256 rows of C depending on the kernel it picks (which moves with the CUs reserved for a collective, with
            # vaw_debug_gemm_tile, ...), the attention backward one per 64 / 128 / 256 query rows by VAW_ATTN_BWD_BIG.  The device
            # table bakes the counts in, so it is rebuilt whenever this backward's counts differ from the ones it was built with
            # (a host-side integer compare per job; folding stale counts would drop or double-count rows silently).
            rows = tuple(int(cp.rows.value) for cp, _ in cps)
            ent = ws.bias_groups.get(key)
            if ent is None or ent[1] != rows:
                jobs = [(cp.buf.data_ptr(), self._g(nm + "bias"), r, cp.N) for (cp, nm), r in zip(cps, rows)]
                ent = ws.bias_groups[key] = (ops.ReduceGroup(jobs, dout.device), rows)
            grp = ent[0]
            grp.launch(beta)

        def flush():
            """One grouped launch for the weight gradients of the blocks in `pending`, then their gradient-ready stages."""
            if not pending:
                return
            key = (pending[0], pending[-1], self._gbase)
            grp = ws.wgrad_groups.get(key)
            if grp is None:
                probs = []
                for l in pending:
                    b, pre = ws.blk[l], f"blocks.{l}."
                    if fp8:     # dy^T [Nw][M] and x^T [Kw][M], with their device scales
                        for name, fdy, fx, Nw, Kw in ((pre + "mlp.fc2.", b["f_dy2"], b["f_a"], D, Dm), (pre + "mlp.fc1.", b["f_dhid"], b["f_xm2"], Dm, D),
                                                      (pre + "attn.proj.", b["f_dy1"], b["f_ao"], D, D), (pre + "attn.qkv.", b["f_dqkv"], b["f_xm"], 3 * D, D)):
                            probs.append((ptr(fdy.qt), ptr(fx.qt), self._g(name + "weight"), Nw, Kw, M, M, Kw, 0.0, 0, ptr(fdy.scale), ptr(fx.scale)))
                        continue
                    for name, dy, x, Nw, Kw in ((pre + "mlp.fc2.", b["dy2"], b["a"], D, Dm), (pre + "mlp.fc1.", b["dDm"], b["xm2"], Dm, D),
                                                (pre + "attn.proj.", b["dy1"], b["ao"], D, D), (pre + "attn.qkv.", b["dqkv"], b["xm"], 3 * D, D)):
                        probs.append((ptr(dy), ptr(x), self._g(name + "weight"), Nw, Kw, Nw, Kw, Kw))
                grp = ws.wgrad_groups[key] = ops.WgradGroup(probs, M, dout.device)
            grp.launch((L.BF8 if self.fp8_grad_format == "e5m2" else L.FP8) if fp8 else dt, beta)
            fold_bias(list(pending), True)
            if hook:
                for l in pending:
                    hook(l + 1)
            pending.clear()

        for l in reversed(range(Lyr)):
            b, pre = ws.blk[l], f"blocks.{l}."
            mo, dmo = mod + 4 * (6 * l * D), dmod + 4 * (6 * l * D)
            xin, xmid = ptr(ws.xres[2 * l]), ptr(ws.xres[2 * l + 1])
            dy2, dhid, dy1, dq = ((ptr(b["dy2"]), ptr(b["dDm"]), ptr(b["dy1"]), ptr(b["dqkv"])) if own_dy
                                  else (ptr(ws.dyb), dDm, ptr(ws.dyb), ptr(ws.dqkv)))
            # MLP branch.  dy2, dgate and the partial column sums of dy2 (fc2's bias gradient) came out of the row kernel that
            # produced this block's incoming residual gradient; bias gradients are folded per group (fold_bias)
            if not defer:
                self._wgrad(dt, pre + "mlp.fc2.", dy2, ptr(b["a"]), D, Dm, M, beta, bias=False)
            fuse_dh = fp8 and ws.d_bwd and self.fp8_fuse_epilogue and M % 64 == 0       # fc2's input-gradient epilogue writes dhid as fp8
            self._linear_dgrad(ws, b, "f_dy2", None if fuse_rows else dy2, pre + "mlp.fc2.", M, D, Dm, b["f_dhid"].epilogue_target() if fuse_dh else dhid, act=2,
                               aux_in=ptr(b["hpre"]), colsum_partial=b["cp_fc1"], **({"out_fp8": b["f_dhid"]} if fuse_dh else {}))
            if not defer:
                self._wgrad(dt, pre + "mlp.fc1.", dhid, ptr(b["xm2"]), Dm, D, M, beta, bias=False)
            self._linear_dgrad(ws, b, "f_dhid", None if (fp8 and fuse_dh) else dhid, pre + "mlp.fc1.", M, Dm, D, dD)
            ln_bwd_and_gate(dD, xmid, ptr(b["mean2"]), ptr(b["rstd2"]), mo + 4 * 4 * D, dres, dmo + 4 * 3 * D, dmo + 4 * 4 * D, (l, "attn"))
            # attention branch (dy1 etc. from the fused row kernel just above)
            if not defer:
                self._wgrad(dt, pre + "attn.proj.", dy1, ptr(b["ao"]), D, D, M, beta, bias=False)
            self._linear_dgrad(ws, b, "f_dy1", None if fuse_rows else dy1, pre + "attn.proj.", M, D, D, ptr(ws.dao))
            q = ptr(b["qkv"])
            qkv_bias_folded = False
            if defer:      # the qkv bias gradient = column sums of dqkv: partial rows from the attention kernels where they offer it
                qkv_bias_folded = ops.attn_bwd_colsum(dt, self._attn_desc(B), q, q + es * D, q + 2 * es * D, ptr(b["ao"]), ptr(ws.dao),
                                                      ptr(b["lse"]), ptr(ws.delta), dq, dq + es * D, dq + 2 * es * D, b["cp_qkv"])
            if not qkv_bias_folded:
                ops.attn_bwd(dt, self._attn_desc(B), q, q + es * D, q + 2 * es * D, ptr(b["ao"]), ptr(ws.dao), ptr(b["lse"]),
                             ptr(ws.delta), dq, dq + es * D, dq + 2 * es * D)
                if defer:
                    ops.colsum(dt, dq, M, 3 * D, 3 * D, b["cp_qkv"].buf.data_ptr(), 0.0, device=dout.device)     # one complete row
                    b["cp_qkv"].rows.value = 1
                else:      # (the per-layer launch takes the bias gradient from its staged dy tiles)
                    self._wgrad(dt, pre + "attn.qkv.", dq, ptr(b["xm"]), 3 * D, D, M, beta)
            self._linear_dgrad(ws, b, "f_dqkv", dq, pre + "attn.qkv.", M, 3 * D, D, dD)
            ln_bwd_and_gate(dD, xin, ptr(b["mean1"]), ptr(b["rstd1"]), mo + 4 * D, dres, dmo, dmo + 4 * D, (l - 1, "mlp") if l else None)
            if defer:
                pending.append(l)
                if l == group_cut:
                    pending.reverse()
                    flush()
            else:
                fold_bias([l], False)
            if hook:
                if not defer:
                    hook(l + 1)
                if l == ada_half:
                    # data-parallel runs: the adaLN rows of blocks >= l (and the final layer's) are final now -- weight and
                    # bias gradients of that part of the packed matrix leave with their own bucket while blocks l-1..0 run
                    r0 = 6 * l * D
                    self._adaln_wgrad(dt, ws, r0, ld - r0, B, beta)
                    hook("ada_hi")
        pending.reverse()
        flush()
        # patch embedding: d(x0) = dres
        if dt == BF16:
            ops.cast_bf16(ws.dres, ws.dD)
            dx0 = dD
        else:
            dx0 = dres
        self._wgrad(dt, "x_embedder.proj.", dx0, ptr(ws.xp), D, self.Kp, M, beta)
        dx = None
        if need_dx:
            ops.gemm(dt, 1, 0, M, self.Kp, D, dx0, D, self._w("x_embedder.proj.weight"), self.Kp, ptr(ws.dxp), self.Kp, out_f32=True)
            dx = torch.empty(B, self.in_channels, H, W, device=dout.device, dtype=torch.float32)
            L.check(lib.vaw_patchify_bwd(ptr(ws.dxp), ptr(dx), B, self.in_channels, H, W, self.patch_size, st), "patchify_bwd")
        # conditioning path: adaLN (all blocks at once, or the rows the early bucket has not covered), label table, timestep MLP
        rows = ld if ada_half is None else 6 * ada_half * D
        dmod_a = self._adaln_wgrad(dt, ws, 0, rows, B, beta)
        ops.gemm(dt, 1, 0, B, D, ld, dmod_a, ld, self._w("blocks.0.adaLN_modulation.1.weight"), D, ptr(ws.dcs), D, out_f32=True)
        L.check(lib.vaw_silu_bwd(ptr(ws.c), ptr(ws.dcs), ptr(ws.dc), B * D, st), "silu_bwd")
        ye = self.y_embedder.embedding_table
        L.check(lib.vaw_embedding_bwd(ptr(ws.dc), ptr(ws.y), self._g("y_embedder.embedding_table.weight"), B, D,
                                      ye.num_embeddings, beta, st), "embedding_bwd")
        if dt == BF16:
            ops.cast_bf16(ws.dc, ws.dc_a)
            dc_a = ptr(ws.dc_a)
        else:
            dc_a = ptr(ws.dc)
        self._wgrad(dt, "t_embedder.mlp.2.", dc_a, ptr(ws.h1s), D, D, ws.Bk, beta)
        ops.gemm(dt, 1, 0, B, D, D, dc_a, D, self._w("t_embedder.mlp.2.weight"), D, ptr(ws.dh1s), D, out_f32=True)
        L.check(lib.vaw_silu_bwd(ptr(ws.h1), ptr(ws.dh1s), ptr(ws.dh1), B * D, st), "silu_bwd")
        if dt == BF16:
            ops.cast_bf16(ws.dh1, ws.dh1_a)
            dh1_a = ptr(ws.dh1_a)
        else:
            dh1_a = ptr(ws.dh1)
        self._wgrad(dt, "t_embedder.mlp.0.", dh1_a, ptr(ws.tfreq), D, 256, ws.Bk, beta)
        self.attach_grads()
        if fp8:
            ws.calib_bwd = True
        if hook:
            hook(0)
        return dx


_PRESETS = {"DiT-S": (384, 12, 6), "DiT-B": (768, 12, 12), "DiT-L": (1024, 24, 16), "DiT-XL": (1152, 28, 16)}


def _make(name):
    hidden, depth, heads = _PRESETS[name]

    def build(image_size, patch_size, in_channels, class_dropout_prob, num_classes, learn_sigma, **kwargs):
        return DiT(image_size=image_size, patch_size=patch_size, in_channels=in_channels, hidden_size=hidden,
                   depth=depth, num_heads=heads, class_dropout_prob=class_dropout_prob, num_classes=num_classes,
                   learn_sigma=learn_sigma, **kwargs)
    build.__name__ = name.replace("-", "_")
    return build


DiT_S, DiT_B, DiT_L, DiT_XL = (_make(n) for n in ("DiT-S", "DiT-B", "DiT-L", "DiT-XL"))
DiT_models = {"DiT-S": DiT_S, "DiT-B": DiT_B, "DiT-L": DiT_L, "DiT-XL": DiT_XL}
