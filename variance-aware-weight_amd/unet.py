"""UNet denoiser on hand-written gfx950 kernels, drop-in for the reference's `models/unet.py` (`UNetModel`
:397-687, `ResBlock` :143-256, `AttentionBlock` :259-306, `QKVAttention[Legacy]` :329-390, presets :921-1021).

Same constructor, parameter names (state_dict keys), seeded initialisation and call protocol
(`model(x, timesteps, y=None) -> Tensor [N, C_out, H, W]`).  Supported topology = what every reference factory
builds (unet.py:936-939): use_scale_shift_norm=True, resblock_updown=True, dropout=0, 2-D; both attention orders.

Data layout in HBM: activations are NHWC -- a [B*H*W, C] row-major matrix in the act dtype -- so that
  * conv3x3 is a GEMM over pixel rows (round 1: explicit im2col patch matrix + the MFMA GEMM; conv weights are
    STORED channels-last, [Co][3][3][Ci] = the GEMM's k-major B operand, and exposed as [Co,Ci,3,3] views),
  * conv1x1 / Conv1d(k=1) are plain GEMMs, and the attention qkv rows are token-major, which is exactly the
    layout of the DiT attention kernels (new order: [3][H][ch]; legacy order: [H][3][ch] -- only strides differ),
  * GroupNorm's channel groups are contiguous inside each pixel row.
All ResBlocks' `emb_layers` Linear weights are packed into one matrix: one GEMM produces every block's FiLM
(scale, shift) from SiLU(emb).  Forward builds a short tape of coarse ops; backward walks it in reverse (no
torch autograd graph inside the model).
"""

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from ._lib import BF16, F32, ptr
from .flat import FlatModule

__all__ = ["UNetModel", "create_unet_model", "UNet_32", "ADM_32", "ADM_64", "ADM_128", "ADM_256", "ADM_512", "UNet_64", "LDM"]


# ---------------------------------------------------------------------------------------------------------
# parameter holders: same nesting and registration order as the reference (=> same keys, same seeded init)
# ---------------------------------------------------------------------------------------------------------
class GroupNorm32(nn.GroupNorm):
    pass


def _zero(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


class TimestepBlock(nn.Module):
    pass


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    pass


class _Resample(nn.Module):
    def __init__(self, up):
        super().__init__()
        self.up = up


class Upsample(nn.Module):
    """nearest x2 (+ conv3x3): reference models/unet.py:81-110 (resblock_updown=False topologies)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=1)


class Downsample(nn.Module):
    """conv3x3 stride 2 (or 2x2 average pool): reference models/unet.py:113-140."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.op = nn.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=1)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)


class ResBlock(TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, up=False, down=False,
                 use_scale_shift_norm=True):
        super().__init__()
        self.use_scale_shift_norm, self.dropout = use_scale_shift_norm, float(dropout)
        self.channels, self.out_channels = channels, out_channels or channels
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(), nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.updown, self.up, self.down = up or down, up, down
        if up or down:
            self.h_upd, self.x_upd = _Resample(up), _Resample(up)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, (2 if use_scale_shift_norm else 1) * self.out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(32, self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        _zero(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)


class _AttnOrder(nn.Module):
    def __init__(self, n_heads, new_order):
        super().__init__()
        self.n_heads, self.new_order = n_heads, new_order


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0
            self.num_heads = channels // num_head_channels
        self.norm = GroupNorm32(32, channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.attention = _AttnOrder(self.num_heads, use_new_attention_order)
        self.proj_out = _zero(nn.Conv1d(channels, channels, 1))


# ---------------------------------------------------------------------------------------------------------
# tape
# ---------------------------------------------------------------------------------------------------------
class _Act:
    """An NHWC activation [B*H*W, C] with its gradient slot."""
    __slots__ = ("t", "B", "H", "W", "C", "grad")

    def __init__(self, t, B, H, W, C):
        self.t, self.B, self.H, self.W, self.C, self.grad = t, B, H, W, C, None

    @property
    def M(self):
        return self.B * self.H * self.W


class UNetModel(FlatModule):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=0, use_checkpoint=False,
                 use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, drop_label_prob=0.0, compute_dtype="bf16"):
        super().__init__()
        if dims != 2:
            raise NotImplementedError("the HIP UNet is 2-D (every reference factory is, unet.py:936-939)")
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError("dropout must be in [0, 1)")
        if model_channels % 32:
            raise ValueError("GroupNorm32 needs channel counts divisible by 32")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_classes, self.drop_label_prob = out_channels, num_classes, drop_label_prob
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = num_res_blocks, attention_resolutions, channel_mult
        self.dropout = dropout
        ted = 512 if in_channels == 4 else model_channels * 4
        self.time_embed_dim = ted
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        if num_classes > 0:
            self.label_emb = nn.Embedding(num_classes + int(drop_label_prob > 0), ted)

        def res(cin, cout, **kw):
            return ResBlock(cin, ted, dropout, out_channels=cout, use_scale_shift_norm=use_scale_shift_norm, **kw)

        def attn(c, heads):
            return AttentionBlock(c, num_heads=heads, num_head_channels=num_head_channels,
                                  use_new_attention_order=use_new_attention_order)

        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, ch, 3, padding=1))])
        chans, ds = [ch], 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, int(mult * model_channels))]
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(res(ch, ch, down=True) if resblock_updown
                                                                 else Downsample(ch, conv_resample, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch, None), attn(ch, num_heads), res(ch, None))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + chans.pop(), int(model_channels * mult))]
                ch = int(model_channels * mult)
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(res(ch, ch, up=True) if resblock_updown else Upsample(ch, conv_resample, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), _zero(nn.Conv2d(input_ch, out_channels, 3, padding=1)))
        # FiLM table: every ResBlock's emb_layers Linear, in module order, becomes one matrix
        for n, m in self.named_modules():
            m._vaw_name = n                      # survives deepcopy (the EMA copy), unlike an id()-keyed table
        off = 0
        for rb in self._resblocks:
            rb.emb_off = off
            off += (2 if rb.use_scale_shift_norm else 1) * rb.out_channels
        self.emb_cols = off
        self.set_compute_dtype(compute_dtype)
        self._anchor = torch.zeros(1, requires_grad=True)
        self._tape, self._tape_gen, self._rec, self._record_next = [], 0, True, True
        # nn.Dropout inside the ResBlocks (--dropout, reference main.py:99): True = the keep masks come from the CPU generator, in the
        # order and shapes the reference's CPU run draws them (parity runs; set by Trainer from args.cpu_rng); False = device RNG
        self.host_dropout_rng = False
        self.grad_ready_hook = None
        self._pend_wgrad, self._wgrad_groups, self._grouped_wgrad = {}, {}, None

    # ---- flat storage -----------------------------------------------------------------------
    @property
    def _resblocks(self):
        return [m for m in self.modules() if isinstance(m, ResBlock)]

    def _flat_groups(self):
        w = [rb._vaw_name + ".emb_layers.1.weight" for rb in self._resblocks]
        b = [rb._vaw_name + ".emb_layers.1.bias" for rb in self._resblocks]
        return [w, b]

    def _flat_channels_last(self, name, p):
        return p.dim() == 4          # conv weights are stored [Co][kh][kw][Ci]

    def set_compute_dtype(self, name):
        name = {"float32": "fp32", "f32": "fp32", "bfloat16": "bf16"}.get(name, name)
        if name not in ("bf16", "fp32"):
            raise ValueError(f"compute_dtype must be 'bf16' or 'fp32', got {name}")
        if name != getattr(self, "compute_dtype", name):
            self.require_fresh_masters("set_compute_dtype()")
        self.compute_dtype = name
        self._dt = BF16 if name == "bf16" else F32

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        self._anchor = fn(self._anchor.detach()).requires_grad_(True)
        return r

    def grad_stage_bounds(self):
        """stage -> range(s) of the flat gradient buffer final when backward reports `stage`.  Backward runs the tape in
        reverse, so the decoder's gradients are complete first: 3 = output_blocks + out, 2 = middle_block, 0 = the rest
        (input_blocks, embeddings, the packed FiLM matrix).  The first two leave while the encoder is still back-propagating."""
        self.ensure_flat()
        names = [n for n, p in self.named_parameters() if p.requires_grad and ".emb_layers." not in n]
        first = lambda prefix: min(self._flat_offsets[n][0] for n in names if n.startswith(prefix))
        mid, dec, n = first("middle_block."), first("output_blocks."), self._flat_n_train
        if not (0 < mid < dec < n):
            return {0: (0, n)}
        # everything from `dec` on is output_blocks.* / out.* and [mid, dec) is middle_block.* (named_parameters order, FiLM packed in front)
        assert all(self._flat_offsets[k][0] >= dec for k in names if k.startswith(("output_blocks.", "out.")))
        assert all(mid <= self._flat_offsets[k][0] < dec for k in names if k.startswith("middle_block."))
        return {3: (dec, n), 2: (mid, dec), 0: (0, mid)}

    # ---- reference surface ------------------------------------------------------------------
    def forward(self, x, timesteps, y=None, force_drop_ids=None, **kwargs):
        assert (y is not None) == (self.num_classes > 0), "must specify y if and only if the model is class-conditional"
        L.need_cuda(x, timesteps, y)
        if self.num_classes > 0:
            if (self.drop_label_prob > 0 and self.training) or force_drop_ids is not None:
                if force_drop_ids is None:
                    drop = torch.rand(y.shape[0]).to(y.device) < self.drop_label_prob     # CPU draw, reference :649
                else:
                    drop = force_drop_ids == 1
                y = torch.where(drop, self.num_classes, y)
            assert y.shape == (x.shape[0],)
        self._record_next = torch.is_grad_enabled()
        return _UNetFn.apply(self._anchor, self, x, timesteps, y)

    @property
    def host_rng_in_forward(self):
        """True when forward() draws random numbers on the host (the CFG label-drop mask, reference models/unet.py:644-653):
        such a step cannot be captured into a hipGraph (the mask would freeze at its capture-time value)."""
        return (self.num_classes > 0 and self.drop_label_prob > 0) or (self.dropout > 0 and self.host_dropout_rng)

    # ---- engine helpers -----------------------------------------------------------------------
    def _w(self, name):
        o, _ = self._flat_offsets[name]
        return self._wbase + o * self._wsize

    def _g(self, name):
        o, _ = self._flat_offsets[name]
        return self._gbase + 4 * o

    def _p32(self, name):
        o, _ = self._flat_offsets[name]
        return self._flat.data_ptr() + 4 * o

    def _new(self, *shape, dtype=None):
        return torch.empty(*shape, device=self._flat.device, dtype=dtype or self._adt)

    # ---- forward ops (each pushes its backward on the tape) -------------------------------------------------
    def _gn(self, a, mod, silu, film=None):
        """GroupNorm32 [+FiLM] [+SiLU].  film = byte address of this block's scale row in the FiLM table."""
        name = mod._vaw_name
        B, HW, C, dt, lib = a.B, a.H * a.W, a.C, self._dt, L.lib()
        y = _Act(self._new(a.M, C), B, a.H, a.W, C)
        mean, rstd = self._new(B * 32, dtype=torch.float32), self._new(B * 32, dtype=torch.float32)
        gam, bet = self._p32(name + ".weight"), self._p32(name + ".bias")
        sc = film
        sh = film + 4 * C if film else 0
        ws = ops.scratch_f32(self._flat.device, lib.vaw_groupnorm_workspace_floats(B, HW, C))
        L.check(lib.vaw_groupnorm_fwd(dt, ptr(a.t), gam, bet, sc or None, sh or None, self.emb_cols, 1 if silu else 0,
                                      ptr(y.t), ptr(mean), ptr(rstd), B, HW, C, 32, 1e-5, ptr(ws), L.stream_ptr()), "groupnorm_fwd")

        def bw():
            dx = self._new(a.M, C)
            dsc = self._demb + (film - self._emb_base) if film else 0
            L.check(lib.vaw_groupnorm_bwd(dt, ptr(y.grad), ptr(a.t), ptr(mean), ptr(rstd), gam, bet, sc or None, sh or None,
                                          self.emb_cols, 1 if silu else 0, ptr(a.grad) if a.grad is not None else None,
                                          ptr(dx), self._g(name + ".weight"), self._g(name + ".bias"), self._beta,
                                          dsc or None, (dsc + 4 * C) if dsc else None, self.emb_cols, B, HW, C, 32,
                                          ptr(ops.scratch_f32(self._flat.device, lib.vaw_groupnorm_workspace_floats(B, HW, C))),
                                          L.stream_ptr()), "groupnorm_bwd")
            a.grad = dx
            y.grad = None
        self._push(bw)
        return y

    def _conv3(self, a, mod, resid=None):
        """conv3x3 pad 1 (+ fused residual add).  GEMM over the im2col patch matrix; weights are stored [Co][9*Ci]."""
        name = mod._vaw_name
        Ci, Co, M, dt, lib = a.C, mod.out_channels, a.M, self._dt, L.lib()
        K = 9 * Ci
        y = _Act(self._new(M, Co), a.B, a.H, a.W, Co)
        wp, bp = self._w(name + ".weight"), self._p32(name + ".bias")
        rp = ptr(resid.t) if resid is not None else None
        geo = (a.B, a.H, a.W, Ci, Co)
        narrow = resid is None and min(Ci, Co) <= 4 and max(Ci, Co) % 8 == 0     # 3-channel stem / output conv
        if narrow:
            rc = lib.vaw_conv3x3_narrow(dt, 0 if Ci <= 4 else 2, ptr(a.t), wp, bp, ptr(y.t), a.B, a.H, a.W, min(Ci, Co), max(Ci, Co),
                                        L.stream_ptr())
            narrow = rc != -3
            if narrow:
                L.check(rc, "conv3x3_narrow")
        if not narrow and not ops.conv3x3(dt, 0, ptr(a.t), None, wp, ptr(y.t), *geo, bias=bp, resid=rp):
            col = self._new(M, K)           # explicit patch matrix: f32 parity mode and channel counts off the 64 grid
            L.check(lib.vaw_im2col3x3(dt, ptr(a.t), ptr(col), a.B, a.H, a.W, Ci, L.stream_ptr()), "im2col")
            ops.gemm(dt, 1, 1, M, Co, K, ptr(col), K, wp, K, ptr(y.t), Co, bias=bp, resid=rp, resid_is_act=True)
            del col

        def bw():
            dy = y.grad
            need_dx = a.grad is not None or self._needs_grad(a)
            gw, gb = self._g(name + ".weight"), self._g(name + ".bias")
            colb = None
            # the bias gradient rides on the implicit weight-gradient GEMM (row sums of the dy tiles it stages anyway)
            fused = min(Ci, Co) > 4 and ops.conv3x3(dt, 2, ptr(dy), ptr(a.t), None, gw, *geo, beta=self._beta, rowsum_a_out=gb,
                                                    rowsum_a_beta=self._beta)
            if not fused:
                ops.colsum(dt, ptr(dy), M, Co, Co, gb, self._beta, device=self._flat.device)
                if min(Ci, Co) <= 4:       # 3-channel stem / output conv: dedicated skinny weight-gradient kernel
                    need = lib.vaw_conv3x3_wgrad_small_workspace_floats(a.B, a.H, a.W, Ci, Co)
                    ws = ops.scratch_f32(self._flat.device, need)
                    L.check(lib.vaw_conv3x3_wgrad_small(dt, ptr(dy), ptr(a.t), gw, self._beta, a.B, a.H, a.W, Ci, Co, ptr(ws),
                                                        ws.numel(), L.stream_ptr()), "conv3x3_wgrad_small")
                else:
                    colb = self._new(M, K)
                    L.check(lib.vaw_im2col3x3(dt, ptr(a.t), ptr(colb), a.B, a.H, a.W, Ci, L.stream_ptr()), "im2col")
                    ops.gemm(dt, 0, 0, Co, K, M, ptr(dy), Co, ptr(colb), K, gw, K, beta=self._beta, out_f32=True)
            if need_dx:
                dx = self._new(M, Ci)
                done = False
                if Co <= 4 and Ci % 8 == 0:
                    rc = lib.vaw_conv3x3_narrow(dt, 1, ptr(dy), wp, None, ptr(dx), a.B, a.H, a.W, Co, Ci, L.stream_ptr())
                    done = rc != -3
                    if done:
                        L.check(rc, "conv3x3_narrow")
                if not done and not ops.conv3x3(dt, 1, ptr(dy), None, wp, ptr(dx), *geo):
                    if colb is None:
                        colb = self._new(M, K)
                    ops.gemm(dt, 1, 0, M, K, Co, ptr(dy), Co, wp, K, ptr(colb), K)     # d(col)
                    L.check(lib.vaw_col2im3x3(dt, ptr(colb), ptr(dx), a.B, a.H, a.W, Ci, L.stream_ptr()), "col2im")
                self._acc(a, dx)
            if resid is not None:
                self._acc(resid, dy)
            y.grad = None
        self._push(bw)
        return y

    def _linear(self, a, wname, bname, Co, resid=None):
        """conv1x1 / Conv1d(k=1): y = a W^T + b (+ resid)."""
        M, Ci, dt = a.M, a.C, self._dt
        y = _Act(self._new(M, Co), a.B, a.H, a.W, Co)
        ops.gemm(dt, 1, 1, M, Co, Ci, ptr(a.t), Ci, self._w(wname), Ci, ptr(y.t), Co, bias=self._p32(bname),
                 resid=ptr(resid.t) if resid is not None else None, resid_is_act=True)

        def bw():
            dy = y.grad
            deferred = self._defer_wgrad(M, Co, Ci, dy, a.t)
            if deferred:
                # deferred: one grouped launch per stage and pixel count (_flush_wgrads); the bias gradient now
                self._pend_wgrad.setdefault(M, []).append((dy, a.t, self._g(wname), Co, Ci))
                ops.colsum(dt, ptr(dy), M, Co, Co, self._g(bname), self._beta, device=self._flat.device)
            else:
                # weight gradient; the bias gradient (row sums of dy^T) rides on the same launch
                ops.gemm(dt, 0, 0, Co, Ci, M, ptr(dy), Co, ptr(a.t), Ci, self._g(wname), Ci, beta=self._beta, out_f32=True,
                         rowsum_a_out=self._g(bname), rowsum_a_beta=self._beta)
            dx = self._new(M, Ci)
            ops.gemm(dt, 1, 0, M, Ci, Co, ptr(dy), Co, self._w(wname), Ci, ptr(dx), Ci)
            self._acc(a, dx)
            if resid is not None:      # (_acc may adopt its argument as resid's gradient buffer: a kept dy is then added to out of place)
                if deferred:
                    dy._vaw_keep = True
                self._acc(resid, dy)
            y.grad = None
        self._push(bw)
        return y

    def _resample(self, a, up):
        dt, lib, C = self._dt, L.lib(), a.C
        Ho, Wo = (a.H * 2, a.W * 2) if up else (a.H // 2, a.W // 2)
        y = _Act(self._new(a.B * Ho * Wo, C), a.B, Ho, Wo, C)
        L.check(lib.vaw_resample2(dt, ptr(a.t), ptr(y.t), a.B, Ho, Wo, C, 1 if up else 0, 1.0 if up else 0.25, L.stream_ptr()), "resample2")

        def bw():
            dx = self._new(a.M, C)
            # nearest^T = 2x2 sum ; avg_pool^T = replicate / 4
            L.check(lib.vaw_resample2(dt, ptr(y.grad), ptr(dx), a.B, a.H, a.W, C, 0 if up else 1, 1.0 if up else 0.25, L.stream_ptr()), "resample2")
            self._acc(a, dx)
            y.grad = None
        self._push(bw)
        return y

    def _cat(self, a, b):
        dt, lib = self._dt, L.lib()
        y = _Act(self._new(a.M, a.C + b.C), a.B, a.H, a.W, a.C + b.C)
        L.check(lib.vaw_concat_channels(dt, ptr(a.t), ptr(b.t), ptr(y.t), a.M, a.C, b.C, 0, L.stream_ptr()), "concat")

        def bw():
            da, db = self._new(a.M, a.C), self._new(b.M, b.C)
            L.check(lib.vaw_concat_channels(dt, ptr(da), ptr(db), ptr(y.grad), a.M, a.C, b.C, 1, L.stream_ptr()), "split")
            self._acc(a, da)
            self._acc(b, db)
            y.grad = None
        self._push(bw)
        return y

    def _attention(self, a, mod):
        name = mod._vaw_name
        C, H, T, B, dt, lib = a.C, mod.num_heads, a.H * a.W, a.B, self._dt, L.lib()
        ch = C // H
        n = self._gn(a, mod.norm, silu=False)
        qkv = self._linear(n, name + ".qkv.weight", name + ".qkv.bias", 3 * C)
        es = self._wsize
        if mod.attention.new_order:      # channels [3][H][ch]
            desc = L.AttnDesc(B, H, T, ch, T * 3 * C, ch, 3 * C, 1, T * C, ch, C, 1, ch ** -0.5)
            ko, vo = es * C, 2 * es * C
        else:                            # legacy: channels [H][3][ch]
            desc = L.AttnDesc(B, H, T, ch, T * 3 * C, 3 * ch, 3 * C, 1, T * C, ch, C, 1, ch ** -0.5)
            ko, vo = es * ch, 2 * es * ch
        o = _Act(self._new(a.M, C), B, a.H, a.W, C)
        lse = self._new(B * H * T, dtype=torch.float32)
        q = ptr(qkv.t)
        ops.attn_fwd(dt, desc, q, q + ko, q + vo, ptr(o.t), ptr(lse))

        def bw():
            dqkv = self._new(a.M, 3 * C)
            delta = self._new(B * H * T, dtype=torch.float32)
            dq = ptr(dqkv)
            ops.attn_bwd(dt, desc, q, q + ko, q + vo, ptr(o.t), ptr(o.grad), ptr(lse), ptr(delta), dq, dq + ko, dq + vo)
            self._acc(qkv, dqkv)
            o.grad = None
        self._push(bw)
        return self._linear(o, name + ".proj_out.weight", name + ".proj_out.bias", C, resid=a)

    def _dropout(self, a, p):
        """nn.Dropout(p) in training mode: a * keep / (1 - p).  The mask is drawn like at::dropout draws it,
        empty_like(input).bernoulli_(1 - p) over the NCHW tensor, from the CPU generator in parity runs."""
        if self.host_dropout_rng:
            keep = torch.empty(a.B, a.C, a.H, a.W).bernoulli_(1 - p).div_(1 - p)
            mask = keep.permute(0, 2, 3, 1).reshape(a.M, a.C).to(self._flat.device, self._adt)
        else:
            mask = (torch.rand(a.M, a.C, device=self._flat.device) < (1 - p)).to(self._adt).mul_(1.0 / (1 - p))
        y = _Act(self._new(a.M, a.C), a.B, a.H, a.W, a.C)
        L.check(L.lib().vaw_mul(self._dt, ptr(a.t), ptr(mask), ptr(y.t), a.M * a.C, L.stream_ptr()), "mul")

        def bw():
            dx = self._new(a.M, a.C)
            L.check(L.lib().vaw_mul(self._dt, ptr(y.grad), ptr(mask), ptr(dx), a.M * a.C, L.stream_ptr()), "mul")
            self._acc(a, dx)
            y.grad = None
        self._push(bw)
        return y

    def _add_emb(self, a, off):
        """h + emb_out (use_scale_shift_norm=False, reference :250-252): the block's row of the packed emb_layers output is added
        to every pixel, in place (a is the fresh output of the preceding conv)."""
        e = self._emb_base + 4 * off
        L.check(L.lib().vaw_rowvec_add(self._dt, ptr(a.t), e, self.emb_cols, a.B, a.H * a.W, a.C, L.stream_ptr()), "rowvec_add")
        y = _Act(a.t, a.B, a.H, a.W, a.C)

        def bw():
            L.check(L.lib().vaw_rowvec_sum(self._dt, ptr(y.grad), self._demb + 4 * off, self.emb_cols, a.B, a.H * a.W, a.C, 0.0,
                                           L.stream_ptr()), "rowvec_sum")
            self._acc(a, y.grad)
            y.grad = None
        self._push(bw)
        return y

    def _subsample(self, a):
        """the even pixels of a: a stride-2 conv3x3 (pad 1) is the stride-1 conv sampled there (4x the arithmetic of a strided
        kernel; Downsample convs exist only in resblock_updown=False topologies, which no reference factory builds)."""
        Ho, Wo = a.H // 2, a.W // 2
        y = _Act(self._new(a.B * Ho * Wo, a.C), a.B, Ho, Wo, a.C)
        L.check(L.lib().vaw_subsample2(self._dt, ptr(a.t), ptr(y.t), a.B, Ho, Wo, a.C, 0, L.stream_ptr()), "subsample2")

        def bw():
            dx = self._new(a.M, a.C)
            L.check(L.lib().vaw_subsample2(self._dt, ptr(y.grad), ptr(dx), a.B, a.H, a.W, a.C, 1, L.stream_ptr()), "subsample2")
            self._acc(a, dx)
            y.grad = None
        self._push(bw)
        return y

    def _resblock(self, x, rb):
        name = rb._vaw_name
        h = self._gn(x, rb.in_layers[0], silu=True)
        if rb.updown:
            h, x = self._resample(h, rb.up), self._resample(x, rb.up)
        h = self._conv3(h, rb.in_layers[2])
        if rb.use_scale_shift_norm:
            h = self._gn(h, rb.out_layers[0], silu=True, film=self._emb_base + 4 * rb.emb_off)
        else:
            h = self._gn(self._add_emb(h, rb.emb_off), rb.out_layers[0], silu=True)
        if rb.dropout > 0 and self.training:
            h = self._dropout(h, rb.dropout)
        if isinstance(rb.skip_connection, nn.Identity):
            skip = x
        elif rb.skip_connection.kernel_size == (1, 1):
            skip = self._linear(x, name + ".skip_connection.weight", name + ".skip_connection.bias", rb.out_channels)
        else:
            skip = self._conv3(x, rb.skip_connection)
        return self._conv3(h, rb.out_layers[3], resid=skip)

    def _run(self, seq, h):
        for layer in seq:
            if isinstance(layer, ResBlock):
                h = self._resblock(h, layer)
            elif isinstance(layer, AttentionBlock):
                h = self._attention(h, layer)
            elif isinstance(layer, nn.Conv2d):
                h = self._conv3(h, layer)
            elif isinstance(layer, Downsample):
                if layer.use_conv:
                    assert h.H % 2 == 0 and h.W % 2 == 0, "Downsample conv: even image sizes"
                    h = self._subsample(self._conv3(h, layer.op))
                else:
                    h = self._resample(h, False)
            elif isinstance(layer, Upsample):
                h = self._resample(h, True)
                if layer.use_conv:
                    h = self._conv3(h, layer.conv)
            else:
                raise TypeError(type(layer))
        return h

    def _stage_done(self, stage):
        self._flush_wgrads()
        if self.grad_ready_hook:
            self.grad_ready_hook(stage)

    # ---- deferred weight gradients of the 1x1 layers (attention qkv / proj_out, skip connections) ---------------------------
    # Per layer these are dy^T x products with K = pixels long and a few hundred rows and columns: a launch of its own must split K
    # over the chip and push every tile through f32 slabs (150-500 TFLOP/s measured, 48 launches per UNet_64 step).  All layers of
    # one backward stage that share a pixel count go into ONE vaw_wgrad_grouped launch instead (whole tiles for every CU, only the
    # last round K-split), as dit.py does for its blocks; the operands are kept alive until then.  bf16 mode only.
    def _defer_wgrad(self, M, Co, Ci, dy, x):
        if self._grouped_wgrad is None:
            import os
            self._grouped_wgrad = os.environ.get("VAW_UNET_GROUPED_WGRAD", "1") != "0"
        if torch.cuda.is_current_stream_capturing():
            return False          # a new group uploads its descriptor table from pinned memory it allocates: not capturable
        if not (self._grouped_wgrad and self._dt == L.BF16 and M % 64 == 0 and Co % 8 == 0 and Ci % 8 == 0 and Co >= 16 and Ci >= 16
                and (dy.data_ptr() | x.data_ptr()) % 16 == 0):
            return False
        # what deferral buys is the gap between a lone split-K launch and the grouped one; what it costs is a column-sum pass over dy
        # (the bias gradient no longer rides on the weight-gradient launch).  A layer with many output tiles AND a long K runs at
        # ~600 TFLOP/s on its own and has a wide dy: ADM_64's qkv at 32 x 32 (1152 x 384, 262 144 pixels: +58 us against a 110 us
        # pass) stays per-layer; measured net of the rule on ADM_64 / UNet_64 in DESIGN.md §6.0
        tiles = ((Co + 255) // 256) * ((Ci + 191) // 192)
        return tiles < 10 or M <= 65536

    def _flush_wgrads(self):
        pend = self._pend_wgrad
        if not pend:
            return
        for K, items in pend.items():
            probs = tuple((ptr(dy), ptr(x), gw, Co, Ci, Co, Ci, Ci) for dy, x, gw, Co, Ci in items)
            grp = self._wgrad_groups.get(probs)
            if grp is None:
                if len(self._wgrad_groups) >= 64:         # (addresses moved: activations are allocated per step)
                    self._wgrad_groups.clear()
                grp = self._wgrad_groups[probs] = ops.WgradGroup(list(probs), K, self._flat.device)
            grp.launch(self._dt, self._beta)
        pend.clear()                                        # the operands may go: the launches are enqueued on this stream

    def _needs_grad(self, a):
        return a is not self._x_act or self._need_dx

    def _acc(self, a, g):
        if a.grad is None:
            a.grad = g
        elif getattr(a.grad, "_vaw_keep", False):     # a deferred weight gradient still reads this buffer: same traffic, new tensor
            a.grad = torch.add(a.grad, g)
        else:
            L.check(L.lib().vaw_add_inplace(self._dt, ptr(a.grad), ptr(g), g.numel(), L.stream_ptr()), "add_inplace")

    # ---- whole-model forward / backward ---------------------------------------------------------------------------
    def _push(self, bw):
        if self._rec:
            self._tape.append(bw)

    _FWD_STATE = ("_s", "_x_act", "_out_act", "_emb_base", "_need_dx")

    def _forward_impl(self, x, t, y, need_dx, record=True):
        """record=False (forward under torch.no_grad(): sampling, evaluation) builds no tape and leaves the tape and the
        saved activations of a pending training forward untouched."""
        if not record:
            keep = {k: getattr(self, k, None) for k in self._FWD_STATE}
            tape = self._tape
            try:
                self._rec = False
                return self._forward_body(x, t, y, False)
            finally:
                self._rec = True
                self._tape = tape
                for k, v in keep.items():
                    setattr(self, k, v)
        self._rec = True
        self._tape_gen += 1
        return self._forward_body(x, t, y, need_dx)

    def _forward_body(self, x, t, y, need_dx):
        self.ensure_flat()
        z = getattr(self, "_zero", None)
        if z is not None:
            z.wait_gathers()         # sharded optimizer: the gathered weights may still be arriving on the collective stream
        dt, lib, st = self._dt, L.lib(), L.stream_ptr()
        self._adt = L.TORCH_DTYPE[dt]
        if dt == BF16:
            sh = self.shadow_bf16()
            self._wbase, self._wsize = sh.data_ptr(), 2
        else:
            self._wbase, self._wsize = self._flat.data_ptr(), 4
        B, C, H, W = x.shape
        assert C == self.in_channels and H == W == self.image_size
        self._need_dx = need_dx
        if self._rec:
            self._tape = []
        ted, mc, f32 = self.time_embed_dim, self.model_channels, torch.float32
        x = x.float().contiguous()
        tf = t.float().contiguous()
        # conditioning: emb = time_embed(sinusoid(t)) [+ label_emb(y)], FiLM table = Linear_all(SiLU(emb))
        s = self._s = {}
        s["tf"] = self._new(B, mc)
        L.check(lib.vaw_timestep_embedding(dt, ptr(tf), ptr(s["tf"]), B, mc, 10000.0, st), "timestep_embedding")
        s["h1"], s["h1s"] = self._new(B, ted, dtype=f32), self._new(B, ted)
        ops.gemm(dt, 1, 1, B, ted, mc, ptr(s["tf"]), mc, self._w("time_embed.0.weight"), mc, ptr(s["h1"]), ted,
                 bias=self._p32("time_embed.0.bias"), out_f32=True)
        L.check(lib.vaw_silu_fwd(dt, ptr(s["h1"]), ptr(s["h1s"]), B * ted, st), "silu")
        s["emb"] = self._new(B, ted, dtype=f32)
        ops.gemm(dt, 1, 1, B, ted, ted, ptr(s["h1s"]), ted, self._w("time_embed.2.weight"), ted, ptr(s["emb"]), ted,
                 bias=self._p32("time_embed.2.bias"), out_f32=True)
        if self.num_classes > 0:
            s["y"] = y.contiguous()
            L.check(lib.vaw_add_embedding(ptr(s["emb"]), self._p32("label_emb.weight"), ptr(s["y"]), ptr(s["emb"]), B, ted,
                                          self.label_emb.num_embeddings, st), "add_embedding")
        s["es"] = self._new(B, ted)
        L.check(lib.vaw_silu_fwd(dt, ptr(s["emb"]), ptr(s["es"]), B * ted, st), "silu")
        s["film"] = self._new(B, self.emb_cols, dtype=f32)
        first = self._resblocks[0]._vaw_name + ".emb_layers.1."
        ops.gemm(dt, 1, 1, B, self.emb_cols, ted, ptr(s["es"]), ted, self._w(first + "weight"), ted, ptr(s["film"]),
                 self.emb_cols, bias=self._p32(first + "bias"), out_f32=True)
        self._emb_base = ptr(s["film"])
        # trunk
        xa = _Act(self._new(B * H * W, C), B, H, W, C)
        L.check(lib.vaw_nchw_to_nhwc(dt, ptr(x), ptr(xa.t), B, C, H * W, st), "nchw_to_nhwc")
        self._x_act = xa
        h, hs = xa, []
        for blk in self.input_blocks:
            h = self._run(blk, h)
            hs.append(h)
        self._push(lambda: self._stage_done(2))      # popped once middle_block has been back-propagated
        h = self._run(self.middle_block, h)
        self._push(lambda: self._stage_done(3))      # ... once output_blocks + out have been
        for blk in self.output_blocks:
            h = self._run(blk, self._cat(h, hs.pop()))
        h = self._conv3(self._gn(h, self.out[0], silu=True), self.out[2])
        self._out_act = h
        out = torch.empty(B, self.out_channels, H, W, device=x.device, dtype=f32)
        L.check(lib.vaw_nhwc_to_nchw(dt, ptr(h.t), ptr(out), B, self.out_channels, H * W, st), "nhwc_to_nchw")
        return out

    def _backward_impl(self, dout):
        dt, lib, st, s = self._dt, L.lib(), L.stream_ptr(), self._s
        B, ted, mc, f32 = dout.shape[0], self.time_embed_dim, self.model_channels, torch.float32
        self._beta = 1.0 if self.grads_live() else 0.0
        self._gbase = self.flat_grads().data_ptr()
        demb_t = torch.zeros(B, self.emb_cols, device=dout.device, dtype=f32)
        self._demb = ptr(demb_t)
        h = self._out_act
        h.grad = self._new(h.M, h.C)
        L.check(lib.vaw_nchw_to_nhwc(dt, ptr(dout), ptr(h.grad), B, h.C, h.H * h.W, st), "nchw_to_nhwc")
        for bw in reversed(self._tape):
            bw()
        self._tape = []
        self._flush_wgrads()
        dx = None
        if self._need_dx:
            xa = self._x_act
            dx = torch.empty(B, xa.C, xa.H, xa.W, device=dout.device, dtype=f32)
            L.check(lib.vaw_nhwc_to_nchw(dt, ptr(xa.grad), ptr(dx), B, xa.C, xa.H * xa.W, st), "nhwc_to_nchw")
        # FiLM table -> emb -> time MLP / label table
        beta = self._beta
        first = self._resblocks[0]._vaw_name + ".emb_layers.1."
        E = self.emb_cols

        def act_copy(t32):
            if dt == F32:
                return t32
            o = self._new(*t32.shape)
            ops.cast_bf16(t32, o)
            return o
        df = act_copy(demb_t)
        ops.gemm(dt, 0, 0, E, ted, B, ptr(df), E, ptr(s["es"]), ted, self._g(first + "weight"), ted, beta=beta, out_f32=True)
        ops.colsum(dt, ptr(df), B, E, E, self._g(first + "bias"), beta, device=self._flat.device)
        des = self._new(B, ted, dtype=f32)
        ops.gemm(dt, 1, 0, B, ted, E, ptr(df), E, self._w(first + "weight"), ted, ptr(des), ted, out_f32=True)
        demb = self._new(B, ted, dtype=f32)
        L.check(lib.vaw_silu_bwd(ptr(s["emb"]), ptr(des), ptr(demb), B * ted, st), "silu_bwd")
        if self.num_classes > 0:
            L.check(lib.vaw_embedding_bwd(ptr(demb), ptr(s["y"]), self._g("label_emb.weight"), B, ted,
                                          self.label_emb.num_embeddings, beta, st), "embedding_bwd")
        da = act_copy(demb)
        ops.gemm(dt, 0, 0, ted, ted, B, ptr(da), ted, ptr(s["h1s"]), ted, self._g("time_embed.2.weight"), ted, beta=beta, out_f32=True)
        ops.colsum(dt, ptr(da), B, ted, ted, self._g("time_embed.2.bias"), beta, device=self._flat.device)
        dh1s = self._new(B, ted, dtype=f32)
        ops.gemm(dt, 1, 0, B, ted, ted, ptr(da), ted, self._w("time_embed.2.weight"), ted, ptr(dh1s), ted, out_f32=True)
        dh1 = self._new(B, ted, dtype=f32)
        L.check(lib.vaw_silu_bwd(ptr(s["h1"]), ptr(dh1s), ptr(dh1), B * ted, st), "silu_bwd")
        db = act_copy(dh1)
        ops.gemm(dt, 0, 0, ted, mc, B, ptr(db), ted, ptr(s["tf"]), mc, self._g("time_embed.0.weight"), mc, beta=beta, out_f32=True)
        ops.colsum(dt, ptr(db), B, ted, ted, self._g("time_embed.0.bias"), beta, device=self._flat.device)
        self.attach_grads()
        self._s, self._x_act, self._out_act = {}, None, None
        if self.grad_ready_hook:
            self.grad_ready_hook(0)
        return dx


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, x, t, y):
        ctx.model = model
        out = model._forward_impl(x, t, y, x.requires_grad, record=model._record_next)
        ctx.gen = model._tape_gen
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.model
        if not m._tape or m._tape_gen != ctx.gen:
            raise L.VawError("UNet backward: the tape of this forward was already consumed, or a later forward of the same "
                             "module replaced it (run sampling / evaluation forwards under torch.no_grad())")
        dx = m._backward_impl(dout.contiguous())
        return torch.zeros_like(m._anchor), None, dx, None, None


_DEFAULT_MULT = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4), 32: (1, 2, 2, 2)}


def create_unet_model(image_size, num_channels, num_res_blocks, channel_mult="", in_channels=3, num_classes=10,
                      learn_sigma=False, class_cond=True, use_checkpoint=False, attention_resolutions="16", num_heads=1,
                      num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0,
                      resblock_updown=True, use_fp16=False, use_new_attention_order=True, drop_label_prob=0.0, **kw):
    if channel_mult == "":
        if image_size not in _DEFAULT_MULT:
            raise ValueError(f"unsupported image size: {image_size}")
        channel_mult = _DEFAULT_MULT[image_size]
    else:
        channel_mult = tuple(int(m) for m in channel_mult.split(","))
    att = tuple(image_size // int(r) for r in attention_resolutions.split(","))
    return UNetModel(image_size=image_size, in_channels=in_channels, model_channels=num_channels,
                     out_channels=(2 * in_channels if learn_sigma else in_channels), num_res_blocks=num_res_blocks,
                     attention_resolutions=att, dropout=dropout, channel_mult=channel_mult,
                     num_classes=(num_classes if class_cond else 0), num_heads=num_heads,
                     num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                     use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                     use_new_attention_order=use_new_attention_order, drop_label_prob=drop_label_prob, **kw)


_PRESETS = {
    "UNet-32": (32, 128, 2, "16,8", 4, -1, "", 3), "ADM-32": (32, 128, 3, "16,8", 1, 32, "", 3),
    "ADM-64": (64, 192, 3, "32,16,8", 1, 64, "", 3), "ADM-128": (128, 256, 2, "32,16,8", 1, 64, "", 3),
    "ADM-256": (256, 256, 2, "32,16,8", 1, 64, "", 3), "ADM-512": (512, 256, 2, "32,16,8", 1, 64, "", 3),
    "UNet-64": (64, 192, 3, "16,8", 4, -1, "1,2,2,2", 3), "LDM": (32, 256, 2, "32,16,8", 1, 32, "1,2,4", 4),
}


def _make(name):
    size, nch, nres, att, heads, hch, mult, in_default = _PRESETS[name]

    def build(num_classes=10, in_channels=in_default, dropout=0, learn_sigma=False, class_cond=True, drop_label_prob=0.0, **kw):
        return create_unet_model(image_size=size, num_channels=nch, num_res_blocks=nres, attention_resolutions=att,
                                 num_heads=heads, num_head_channels=hch, channel_mult=mult, num_classes=num_classes,
                                 dropout=dropout, in_channels=in_channels, drop_label_prob=drop_label_prob,
                                 learn_sigma=learn_sigma, class_cond=class_cond, **kw)
    build.__name__ = name.replace("-", "_")
    return build


UNet_32, ADM_32, ADM_64, ADM_128, ADM_256, ADM_512, UNet_64, LDM = (_make(n) for n in _PRESETS)
UNet_models = {"UNet-32": UNet_32, "ADM-32": ADM_32, "ADM-64": ADM_64, "ADM-128": ADM_128, "ADM-256": ADM_256,
               "ADM-512": ADM_512, "UNet-64": UNet_64, "LDM": LDM}
