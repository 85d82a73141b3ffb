"""Oracle (test infrastructure): restatement of the three timm==0.9.2 classes the
reference DiT imports (/root/reference/models/dit.py:17, requirements.txt:110).

PARITY UNPINNED at this boundary: timm is a third-party dependency that is
neither vendored in /root/reference nor installed in this image, and the
reference holds no test or fixture for it.  What is restated is timm's
published algorithm for `timm.models.vision_transformer.{Attention, Mlp}` and
`timm.layers.PatchEmbed` at the call sites dit.py:126,130,192 (no qk-norm, no
dropout, norm_layer=None):
  Attention : qkv = Linear(D,3D,bias) -> view [B,N,3,H,hd] -> q,k,v [B,H,N,hd]
              -> softmax(q*hd^-0.5 @ k^T) @ v -> [B,N,D] -> Linear(D,D)
  Mlp       : fc1 -> act -> fc2
  PatchEmbed: Conv2d(C,D,k=p,s=p,bias) -> flatten(2).transpose(1,2)
Attention and Mlp are pinned against the reference's OWN statements of the same
math -- models/uvit.py:55-93 ('math' and 'flash' modes) and tools/timm.py:96-112,
run unmodified by tests/golden/make_goldens.py::gen_uvit_anchor -- in
tests/test_oracle_goldens.py::test_timm_restatement_vs_reference_uvit_attention_and_mlp
(outputs and input gradients, same state_dict keys).  What stays unpinned is
only "timm 0.9.2 == the reference's uvit statement" (and PatchEmbed's Conv2d).
Parameter names (`qkv`, `proj`, `fc1`, `fc2`, `proj`) follow timm so reference
checkpoints keep their state_dict keys.
"""
import torch
import torch.nn as nn


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
        p = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((p @ v).transpose(1, 2).reshape(B, N, C))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        hidden_features = hidden_features or in_features
        out_features = out_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)
