"""Vision stacks around AttentionFactory.build_attention (channels-last token grids [B, H, W, C]).

State-dict keys follow the reference models so that their checkpoints load with strict=True:
DeiTStack <-> EfficientTransformer (vit/models/efficient_vit.py:122-233), PvTStack <->
PyramidVisionTransformerV2 (vit/models/pvt_legacy.py:187-268).  tests/test_harness.py checks the key /
shape tables against tests/golden/model_keys.json (dumped from the reference by
tests/golden/gen_model_keys.py)."""
import math
from functools import partial

import torch
import torch.nn as nn

from efficient_attention import AttentionFactory


def _init(m):
    """trunc-normal(.02) Linear weights, zero biases, unit LayerNorm, fan-out normal convolutions
    (efficient_vit.py:195-202, pvt_legacy.py:219-232)."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


def _init_conv(m):
    if isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            m.bias.data.zero_()


class StochasticDepth(nn.Module):
    """Per-sample residual-branch drop (timm's DropPath, used at efficient_vit.py:111-119)."""

    def __init__(self, p):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0. or not self.training:
            return x
        keep = 1.0 - self.p
        shape = (x.shape[0],) + (1,) * (x.dim() - 1)
        return x * x.new_empty(shape).bernoulli_(keep).div_(keep)


class FeedForward(nn.Module):
    """fc1 -> act (optionally gated) -> fc2 (vit/models/model_utils.py:11-45)."""

    def __init__(self, dim, hidden, drop=0., use_glu=False):
        super().__init__()
        if use_glu:
            hidden = int(hidden * 2 // 3)
        self.use_glu = use_glu
        self.fc1 = nn.Linear(dim, hidden * (2 if use_glu else 1))
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        if self.use_glu:
            x, v = self.fc1(x).chunk(2, dim=-1)
            x = self.act(x) * v
        else:
            x = self.act(self.fc1(x))
        return self.drop(self.fc2(self.drop(x)))


class PatchStem(nn.Module):
    """Non-overlapping patchify convolution -> channels-last grid (efficient_vit.py:35-86, 'default' stem)."""

    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.new_H = self.new_W = img_size // patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        assert x.shape[-2:] == self.img_size, (x.shape, self.img_size)
        return self.proj(x).permute(0, 2, 3, 1)


class EncoderBlock(nn.Module):
    """x + attn(norm1(x)); x + mlp(norm2(x))  (efficient_vit.py:97-119)."""

    def __init__(self, attn_name, attn_args, dim, mlp_ratio, drop_path, drop_rate=0., norm_layer=nn.LayerNorm,
                 use_glu=False):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = AttentionFactory.build_attention(attn_name=attn_name, attn_args=attn_args)
        self.drop_path = StochasticDepth(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = FeedForward(dim, int(dim * mlp_ratio), drop=drop_rate, use_glu=use_glu)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class DeiTStack(nn.Module):
    """DeiT-style classifier on a token grid: stem, learned position embedding, `depth` blocks, LayerNorm,
    mean pool, Linear head (efficient_vit.py:122-233)."""

    def __init__(self, attn_name, attn_specific_args, img_size=224, patch_size=16, in_chans=3, num_classes=1000,
                 embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0.1, use_glu=False, use_pos_emb=True):
        super().__init__()
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_classes, self.embed_dim, self.depth = num_classes, embed_dim, depth
        self.patch_embed = PatchStem(img_size, patch_size, in_chans, embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.use_pos_emb = use_pos_emb
        if use_pos_emb:
            self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.new_H, self.patch_embed.new_W, embed_dim))
            self.pos_drop = nn.Dropout(p=drop_rate)
            nn.init.trunc_normal_(self.pos_embed, std=.02)
        self.norm_before_pooling = norm_layer(embed_dim)
        attn_args = dict(attn_specific_args)
        attn_args.update(dim=embed_dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop_rate,
                         proj_drop=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            EncoderBlock(attn_name, attn_args, embed_dim, mlp_ratio, dpr[i], drop_rate, norm_layer, use_glu)
            for i in range(depth)])
        self.apply(_init)

    def forward_features(self, x):
        x = self.patch_embed(x)
        if self.use_pos_emb:
            x = self.pos_drop(x + self.pos_embed)
        B, H, W, C = x.shape
        for blk in self.blocks:
            x = blk(x)
        return self.norm_before_pooling(x.reshape(B, H * W, C)).mean(1)

    def forward(self, x):
        return self.head(self.forward_features(x))


def deit_tiny(attn_name, attn_specific_args, patch_size, **kw):
    """evit_tiny_p16 / evit_tiny_p8 (efficient_vit.py:261-272,300-311): 192 channels, 3 heads, 12 blocks."""
    return DeiTStack(attn_name, attn_specific_args, patch_size=patch_size, embed_dim=192, num_heads=3, depth=12, **kw)


# ------------------------------------------------------------------------------------------
# PvT-v2
# ------------------------------------------------------------------------------------------
class OverlapStem(nn.Module):
    """Strided overlapping convolution + LayerNorm -> channels-last grid (pvt_legacy.py:135-184)."""

    def __init__(self, patch_size, stride, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x):
        return self.norm(self.proj(x).permute(0, 2, 3, 1))


class DepthwiseConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x):                                  # [B, H, W, C]
        return self.dwconv(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)


class ConvFeedForward(nn.Module):
    """fc1 -> depthwise 3x3 -> GELU -> fc2 (pvt_legacy.py:25-63)."""

    def __init__(self, dim, hidden, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = DepthwiseConv(hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        x = self.drop(self.act(self.dwconv(self.fc1(x))))
        return self.drop(self.fc2(x))


class StageAttention(nn.Module):
    """sr_ratio > 1: the configured efficient attention (LARA's kernel_size <- sr_ratio); sr_ratio == 1:
    softmax (pvt_legacy.py:66-93)."""

    def __init__(self, attn_name, attn_specific_args, dim, num_heads, qkv_bias, attn_drop, proj_drop, sr_ratio):
        super().__init__()
        base = dict(dim=dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=proj_drop)
        if sr_ratio > 1:
            args = dict(attn_specific_args)
            args.update(base)
            if "kernel_size" in args:
                args["kernel_size"] = sr_ratio
            self.attn_fn = AttentionFactory.build_attention(attn_name=attn_name, attn_args=args)
        else:
            self.attn_fn = AttentionFactory.build_attention(attn_name="softmax", attn_args=base)

    def forward(self, x):
        return self.attn_fn(x)


class PvTBlock(nn.Module):
    def __init__(self, attn_name, attn_specific_args, dim, num_heads, mlp_ratio, qkv_bias, drop, attn_drop,
                 drop_path, norm_layer, sr_ratio):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = StageAttention(attn_name, attn_specific_args, dim, num_heads, qkv_bias, attn_drop, drop, sr_ratio)
        self.drop_path = StochasticDepth(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = ConvFeedForward(dim, int(dim * mlp_ratio), drop=drop)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class PvTStack(nn.Module):
    """Four stages of [overlapping stem, blocks, LayerNorm]; mean pool; Linear head (pvt_legacy.py:187-268)."""

    def __init__(self, attn_name, attn_specific_args, img_size=224, in_chans=3, num_classes=1000,
                 embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), mlp_ratios=(8, 8, 4, 4),
                 depths=(3, 4, 6, 3), sr_ratios=(8, 4, 2, 1), qkv_bias=True, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0.1):
        super().__init__()
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_classes, self.depths, self.num_stages = num_classes, tuple(depths), 4
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for i in range(4):
            stem = OverlapStem(7 if i == 0 else 3, 4 if i == 0 else 2, in_chans if i == 0 else embed_dims[i - 1],
                               embed_dims[i])
            blocks = nn.ModuleList([
                PvTBlock(attn_name, attn_specific_args, embed_dims[i], num_heads[i], mlp_ratios[i], qkv_bias, drop_rate,
                         attn_drop_rate, dpr[cur + j], norm_layer, sr_ratios[i]) for j in range(depths[i])])
            cur += depths[i]
            setattr(self, "patch_embed%d" % (i + 1), stem)
            setattr(self, "block%d" % (i + 1), blocks)
            setattr(self, "norm%d" % (i + 1), norm_layer(embed_dims[i]))
        self.head = nn.Linear(embed_dims[3], num_classes) if num_classes > 0 else nn.Identity()
        self.apply(_init)
        self.apply(_init_conv)

    def forward_features(self, x):
        for i in range(4):
            x = getattr(self, "patch_embed%d" % (i + 1))(x)
            for blk in getattr(self, "block%d" % (i + 1)):
                x = blk(x)
            x = getattr(self, "norm%d" % (i + 1))(x)
            if i != 3:
                x = x.permute(0, 3, 1, 2).contiguous()
        return x.mean(dim=(1, 2))

    def forward(self, x):
        return self.head(self.forward_features(x))


def pvt_b2(attn_name, attn_specific_args, **kw):
    """pvt_small = PvT-v2-b2 (pvt_legacy.py:349-359): depths 3-4-6-3."""
    return PvTStack(attn_name, attn_specific_args, depths=(3, 4, 6, 3), **kw)
