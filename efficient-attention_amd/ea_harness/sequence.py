"""A fairseq-free Time x Batch x Channel encoder around AttentionFactory.build_attention.

TimeFirstSelfAttention is the adapter of fairseq/fairseq/modules/efficient_attention.py:43-132: it
builds the attention from (attn_name, attn_args) with dim / num_heads / qkv_bias / attn_drop / proj_drop
filled in, takes `query [T, B, C]` and `key_padding_mask [B, T]` (1 = pad), calls the module batch-first
and hands back `[T, B, C]`.  EncoderLayer is the post-norm residual layer of transformer_wmt_en_de
(self-attention, LayerNorm, ReLU feed-forward, LayerNorm).  Parameter names under `self_attn.attn.*`
are the attention module's own, as in the reference adapter."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from efficient_attention import AttentionFactory


class TimeFirstSelfAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, attn_name, attn_args=None):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        args = dict(attn_args or {})
        args.update(dim=embed_dim, num_heads=num_heads, qkv_bias=True, attn_drop=0.0, proj_drop=0.0)
        self.attn = AttentionFactory.build_attention(attn_name=attn_name, attn_args=args)

    def forward(self, query, key=None, value=None, key_padding_mask=None, attn_mask=None):
        T, B, C = query.shape
        assert C == self.embed_dim and attn_mask is None
        if key_padding_mask is not None:
            assert tuple(key_padding_mask.shape) == (B, T)
        out = self.attn(query.transpose(0, 1), key_padding_mask)            # [B, T, C]
        return out.transpose(0, 1).contiguous().view(T, B, C), None


class EncoderLayer(nn.Module):
    def __init__(self, embed_dim, ffn_dim, num_heads, attn_name, attn_args, dropout=0.1, normalize_before=False):
        super().__init__()
        self.self_attn = TimeFirstSelfAttention(embed_dim, num_heads, attn_name, attn_args)
        self.self_attn_layer_norm = nn.LayerNorm(embed_dim)
        self.fc1 = nn.Linear(embed_dim, ffn_dim)
        self.fc2 = nn.Linear(ffn_dim, embed_dim)
        self.final_layer_norm = nn.LayerNorm(embed_dim)
        self.dropout = nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def forward(self, x, key_padding_mask=None):
        res = x
        if self.normalize_before:
            x = self.self_attn_layer_norm(x)
        x, _ = self.self_attn(x, x, x, key_padding_mask=key_padding_mask)
        x = res + self.dropout(x)
        if not self.normalize_before:
            x = self.self_attn_layer_norm(x)
        res = x
        if self.normalize_before:
            x = self.final_layer_norm(x)
        x = self.fc2(self.dropout(F.relu(self.fc1(x))))
        x = res + self.dropout(x)
        if not self.normalize_before:
            x = self.final_layer_norm(x)
        return x


class EncoderStack(nn.Module):
    """Token embedding (scaled) + sinusoidal positions -> `layers` EncoderLayers -> [T, B, C]; a tied
    output projection turns it into token logits so that a training step has a loss."""

    def __init__(self, vocab, embed_dim, ffn_dim, num_heads, layers, attn_name, attn_args=None, dropout=0.1,
                 max_positions=4096, pad_idx=1):
        super().__init__()
        self.embed_dim, self.pad_idx = embed_dim, pad_idx
        self.embed_tokens = nn.Embedding(vocab, embed_dim, padding_idx=pad_idx)
        nn.init.normal_(self.embed_tokens.weight, mean=0, std=embed_dim ** -0.5)
        nn.init.constant_(self.embed_tokens.weight[pad_idx], 0)
        self.register_buffer("positions", self._sinusoid(max_positions, embed_dim), persistent=False)
        self.dropout = nn.Dropout(dropout)
        self.layers = nn.ModuleList([EncoderLayer(embed_dim, ffn_dim, num_heads, attn_name, attn_args, dropout)
                                     for _ in range(layers)])

    @staticmethod
    def _sinusoid(n, dim):
        half = dim // 2
        freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(torch.log(torch.tensor(10000.0)) / (half - 1)))
        ang = torch.arange(n, dtype=torch.float32).unsqueeze(1) * freq.unsqueeze(0)
        return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)

    def forward(self, tokens, key_padding_mask=None):                      # tokens [B, T] int64, mask [B, T] (1 = pad)
        B, T = tokens.shape
        x = self.embed_tokens(tokens) * (self.embed_dim ** 0.5) + self.positions[:T].to(self.embed_tokens.weight.dtype)
        x = self.dropout(x).transpose(0, 1)                                 # [T, B, C]
        for layer in self.layers:
            x = layer(x, key_padding_mask)
        return x

    def logits(self, x):                                                    # [T, B, C] -> [T, B, vocab]
        return F.linear(x, self.embed_tokens.weight)


def wmt_en_de_encoder(attn_name, attn_args=None, vocab=32768, **kw):
    """Encoder half of transformer_wmt_en_de (512 / 2048 / 8 heads / 6 layers, post-norm)."""
    return EncoderStack(vocab, 512, 2048, 8, 6, attn_name, attn_args, **kw)
