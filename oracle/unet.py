"""Oracle (test infrastructure): CPU restatement of the reference UNet denoiser.

Follows /root/reference/models/unet.py: Upsample :81-110, Downsample :113-140,
ResBlock :143-256, AttentionBlock :259-306, QKVAttentionLegacy :329-355,
QKVAttention :362-390, UNetModel :397-687, create_unet_model + presets :921-1021;
and /root/reference/tools/nn.py: GroupNorm32 :17-19, zero_module :68-74,
timestep_embedding :103-121.  The reference's unconditional activation
checkpointing of AttentionBlock (:297) and its autocast() wrapper (:302) change
memory/precision policy, not fp32 arithmetic, and are not reproduced.
Module nesting mirrors the reference so state_dict keys and seeded
initialisation are identical.  Pinned by tests/golden/unet_*.pt.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .dit import sinusoid as timestep_embedding


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


def normalization(ch):
    return GroupNorm32(32, ch)


def zero_module(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


class TimestepBlock(nn.Module):
    pass


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def forward(self, x, emb):
        for layer in self:
            x = layer(x, emb) if isinstance(layer, TimestepBlock) else layer(x)
        return x


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, out_channels=None):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.conv = nn.Conv2d(channels, self.out_channels, 3, padding=1)

    def forward(self, x):
        assert x.shape[1] == self.channels
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return self.conv(x) if self.use_conv else x


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, out_channels=None):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        if use_conv:
            self.op = nn.Conv2d(channels, self.out_channels, 3, stride=2, padding=1)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return self.op(x)


class ResBlock(TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, up=False, down=False):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_scale_shift_norm = use_scale_shift_norm
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.updown = up or down
        if up:
            self.h_upd, self.x_upd = Upsample(channels, False), Upsample(channels, False)
        elif down:
            self.h_upd, self.x_upd = Downsample(channels, False), Downsample(channels, False)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(
            nn.SiLU(), nn.Linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)

    def forward(self, x, emb):
        if self.updown:
            h = self.in_layers[1](self.in_layers[0](x))
            h, x = self.h_upd(h), self.x_upd(x)
            h = self.in_layers[2](h)
        else:
            h = self.in_layers(x)
        e = self.emb_layers(emb).type(h.dtype)[..., None, None]
        if self.use_scale_shift_norm:
            scale, shift = torch.chunk(e, 2, dim=1)
            h = self.out_layers[0](h) * (1 + scale) + shift
            h = self.out_layers[1:](h)
        else:
            h = self.out_layers(h + e)
        return self.skip_connection(x) + h


def _attend(q, k, v, ch):
    """q,k,v: [B*H, ch, T]; scale ch^-1/4 on q and on k; fp32 softmax over s."""
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * s, k * s)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    return torch.einsum("bts,bcs->bct", w, v)


class QKVAttentionLegacy(nn.Module):
    """heads split first, then q|k|v inside each head (unet.py:348)."""

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads

    def forward(self, qkv):
        bs, width, length = qkv.shape
        assert width % (3 * self.n_heads) == 0
        ch = width // (3 * self.n_heads)
        q, k, v = qkv.reshape(bs * self.n_heads, ch * 3, length).split(ch, dim=1)
        return _attend(q, k, v, ch).reshape(bs, -1, length)


class QKVAttention(nn.Module):
    """q|k|v split first, then heads (unet.py:381-389)."""

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads

    def forward(self, qkv):
        bs, width, length = qkv.shape
        assert width % (3 * self.n_heads) == 0
        ch = width // (3 * self.n_heads)
        q, k, v = (z.reshape(bs * self.n_heads, ch, length) for z in qkv.chunk(3, dim=1))
        return _attend(q, k, v, ch).reshape(bs, -1, length)


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0
            self.num_heads = channels // num_head_channels
        self.norm = normalization(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.attention = (QKVAttention if use_new_attention_order else QKVAttentionLegacy)(self.num_heads)
        self.proj_out = zero_module(nn.Conv1d(channels, channels, 1))

    def forward(self, x):
        b, c, *spatial = x.shape
        x = x.reshape(b, c, -1)
        h = self.proj_out(self.attention(self.qkv(self.norm(x))))
        return (x + h).reshape(b, c, *spatial)


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=0,
                 use_checkpoint=False, use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 drop_label_prob=0.0):
        super().__init__()
        assert dims == 2
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_classes, self.drop_label_prob = out_channels, num_classes, drop_label_prob
        ted = 512 if in_channels == 4 else model_channels * 4
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
                self.input_blocks.append(TimestepEmbedSequential(
                    res(ch, ch, down=True) if resblock_updown else Downsample(ch, conv_resample, out_channels=ch)))
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
                    layers.append(res(ch, ch, up=True) if resblock_updown
                                  else Upsample(ch, conv_resample, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(),
                                 zero_module(nn.Conv2d(input_ch, out_channels, 3, padding=1)))

    def forward(self, x, timesteps, y=None, force_drop_ids=None, **kwargs):
        assert (y is not None) == (self.num_classes > 0), "must specify y if and only if the model is class-conditional"
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels))
        if self.num_classes > 0:
            if (self.drop_label_prob > 0 and self.training) or force_drop_ids is not None:
                if force_drop_ids is None:
                    drop = torch.rand(y.shape[0]).to(y.device) < self.drop_label_prob   # CPU draw, reference :649
                else:
                    drop = force_drop_ids == 1
                y = torch.where(drop, self.num_classes, y)
            assert y.shape == (x.shape[0],)
            emb = emb + self.label_emb(y)
        hs, h = [], x
        for m in self.input_blocks:
            h = m(h, emb)
            hs.append(h)
        h = self.middle_block(h, emb)
        for m in self.output_blocks:
            h = m(torch.cat([h, hs.pop()], dim=1), emb)
        return self.out(h)


_DEFAULT_MULT = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4),
                 32: (1, 2, 2, 2)}


def create_unet_model(image_size, num_channels, num_res_blocks, channel_mult="", in_channels=3, num_classes=10,
                      learn_sigma=False, class_cond=True, use_checkpoint=False, attention_resolutions="16",
                      num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0,
                      resblock_updown=True, use_fp16=False, use_new_attention_order=True, drop_label_prob=0.0):
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
                     use_new_attention_order=use_new_attention_order, drop_label_prob=drop_label_prob)


# name: (image_size, num_channels, num_res_blocks, attention_resolutions, num_heads, num_head_channels, channel_mult, in_ch)
_PRESETS = {
    "UNet-32": (32, 128, 2, "16,8", 4, -1, "", 3),
    "ADM-32": (32, 128, 3, "16,8", 1, 32, "", 3),
    "ADM-64": (64, 192, 3, "32,16,8", 1, 64, "", 3),
    "ADM-128": (128, 256, 2, "32,16,8", 1, 64, "", 3),
    "ADM-256": (256, 256, 2, "32,16,8", 1, 64, "", 3),
    "ADM-512": (512, 256, 2, "32,16,8", 1, 64, "", 3),
    "UNet-64": (64, 192, 3, "16,8", 4, -1, "1,2,2,2", 3),
    "LDM": (32, 256, 2, "32,16,8", 1, 32, "1,2,4", 4),
}


def _make(name):
    size, nch, nres, att, heads, hch, mult, in_default = _PRESETS[name]

    def build(num_classes=10, in_channels=in_default, dropout=0, learn_sigma=False, class_cond=True,
              drop_label_prob=0.0, **kw):
        return create_unet_model(image_size=size, num_channels=nch, num_res_blocks=nres, attention_resolutions=att,
                                 num_heads=heads, num_head_channels=hch, channel_mult=mult, num_classes=num_classes,
                                 dropout=dropout, in_channels=in_channels, drop_label_prob=drop_label_prob,
                                 learn_sigma=learn_sigma, class_cond=class_cond, **kw)
    return build


UNet_32, ADM_32, ADM_64, ADM_128, ADM_256, ADM_512, UNet_64, LDM = (_make(n) for n in _PRESETS)
