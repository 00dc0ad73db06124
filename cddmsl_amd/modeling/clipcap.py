"""ClipCap vision-to-language mapper (frozen) -- detectron2/modeling/backbone/clipcap/clipcap.py:39-163,714-719.

Only ``TransformerMapper`` is built (``ClipCaptionModel.clip_project``, engine/train_loop.py:281-288); GPT-2 is
never constructed (it is off the hot path and needs a network fetch).  Parameter names match ``clip_project.*``.
The linears (31 M + 38 M frozen parameters, 3.13 GMAC/sample) run on the HIP MFMA GEMM with input-gradient only;
LayerNorm is a fused HIP kernel (f32 residual stream in, GEMM operand out); the 80-token softmax(QK^T)V core is one fused
bf16 MFMA kernel per direction (csrc/attn_small.hip) on the throughput path and fp32 torch ops on the exact-f32 parity path.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import layers


class _Lin(nn.Module):
    def __init__(self, i, o, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.empty(o)) if bias else None
        self._pw = None

    def forward(self, x2d, relu=False, out_f32=True, residual=None):
        if self._pw is None or self._pw.param is not self.weight:
            self._pw = layers.PreparedWeight(self.weight, None, frozen=True)
        return layers.linear(x2d, self._pw, self.bias, relu=relu, out_f32=out_f32, train_w=False, residual=residual)


class MlpTransformer(nn.Module):
    def __init__(self, in_dim, h_dim):
        super().__init__()
        self.fc1, self.fc2 = _Lin(in_dim, h_dim), _Lin(h_dim, in_dim)

    def forward(self, y, residual):
        """y [M, in_dim] bf16, residual [M, in_dim] f32 -> residual + fc2(relu(fc1(y))) f32 (frozen weights)"""
        for lin in (self.fc1, self.fc2):
            if lin._pw is None or lin._pw.param is not lin.weight:
                lin._pw = layers.PreparedWeight(lin.weight, None, frozen=True)
        return layers.frozen_mlp(y, self.fc1._pw, self.fc1.bias, self.fc2._pw, self.fc2.bias, residual)


class MultiHeadAttention(nn.Module):
    def __init__(self, dim, num_heads, bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.to_queries = _Lin(dim, dim, bias)
        self.to_keys_values = _Lin(dim, dim * 2, bias)
        self.project = _Lin(dim, dim)
        self._qkv = None          # (key, fused frozen weight [3 dim, dim], its PreparedWeight)

    def qkv_weight(self):
        """to_queries and to_keys_values (bias-free, frozen) as ONE [3 dim, dim] GEMM operand, rebuilt when either changes"""
        wq, wkv = self.to_queries.weight, self.to_keys_values.weight
        key = (wq.data_ptr(), wq._version, wkv.data_ptr(), wkv._version)
        if self._qkv is None or self._qkv[0] != key:
            w = torch.nn.Parameter(torch.cat([wq.detach(), wkv.detach()], dim=0), requires_grad=False)
            self._qkv = (key, w, layers.PreparedWeight(w, None, frozen=True))
        return self._qkv[2]


class TransformerLayer(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=2.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = MultiHeadAttention(dim, num_heads, bias=False)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MlpTransformer(dim, int(dim * mlp_ratio))


class Transformer(nn.Module):
    def __init__(self, dim, num_heads, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([TransformerLayer(dim, num_heads) for _ in range(num_layers)])


class TransformerMapper(nn.Module):
    def __init__(self, dim_clip=1024, dim_embedding=768, prefix_length=40, clip_length=40, num_layers=8,
                 compute_dtype=torch.bfloat16):
        super().__init__()
        self.clip_length, self.dim, self.compute_dtype = clip_length, dim_embedding, compute_dtype
        self.transformer = Transformer(dim_embedding, 8, num_layers)
        self.linear = _Lin(dim_clip, clip_length * dim_embedding)
        self.prefix_const = nn.Parameter(torch.randn(prefix_length, dim_embedding))
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x, last_only=False):
        """x [N, dim_clip] f32 -> [N, prefix_length, dim] f32   (clipcap.py:151-155).
        ``last_only=True`` returns only the LAST mapped token [N, dim] -- what ``v2l`` keeps (clipcap.py:714-719).  The other
        79 outputs of the final layer are never read, so that layer then forms queries, the attention output, the
        projection and the MLP for the last token only (keys / values still come from all tokens); same values, ~9 % less
        mapper work."""
        T, d = self.compute_dtype, self.dim
        n = x.shape[0]
        h = self.linear(x.to(T)).view(n, self.clip_length, d)
        h = torch.cat((h, self.prefix_const.unsqueeze(0).expand(n, -1, -1)), dim=1)       # [n, 80, d] f32
        t = h.shape[1]
        nl = len(self.transformer.layers)
        for li, lyr in enumerate(self.transformer.layers):
            a = lyr.attn
            H = a.num_heads
            if last_only and li == nl - 1:
                return self._last_token_layer(lyr, h, T)
            if T == torch.bfloat16 and d // H == 96 and t <= 96:
                # throughput path: projections emit bf16, one fused attention kernel per direction, heads stay column blocks;
                # LayerNorm hands back its input for the residual add so that the backward accumulates in one kernel
                y, hs = layers.layer_norm_skip(h.view(n * t, d), lyr.norm1.weight, lyr.norm1.bias, T)
                if a.to_queries.bias is None and a.to_keys_values.bias is None:
                    qkv = layers.linear(y, a.qkv_weight(), None, out_f32=False, train_w=False)     # one projection GEMM
                    o = layers.small_attention_qkv(qkv, t, H, a.scale)
                else:
                    o = layers.small_attention(a.to_queries(y, out_f32=False), a.to_keys_values(y, out_f32=False), t, H, a.scale)
                h = a.project(o, residual=hs)                 # the f32 residual adds ride in the GEMM epilogues
                y, hs = layers.layer_norm_skip(h, lyr.norm2.weight, lyr.norm2.bias, T)
                h = lyr.mlp(y, hs).view(n, t, d)              # fc2(relu(fc1 y)) + hs as one node (ReLU backward fused)
                continue
            y = layers.layer_norm(h.view(n * t, d), lyr.norm1.weight, lyr.norm1.bias, T)
            # exact-f32 parity path: the same arithmetic on torch ops (the fused kernel is bf16-only)
            q = a.to_queries(y).view(n, t, H, d // H).permute(0, 2, 1, 3)
            kv = a.to_keys_values(y).view(n, t, 2, H, d // H)
            k, v = kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)
            att = torch.softmax((q @ k.transpose(-1, -2)) * a.scale, dim=-1)
            o = (att @ v).permute(0, 2, 1, 3).reshape(n * t, d)
            h = h + a.project(o.to(T)).view(n, t, d)
            y = layers.layer_norm(h.view(n * t, d), lyr.norm2.weight, lyr.norm2.bias, T)
            y = lyr.mlp.fc2(lyr.mlp.fc1(y, relu=True, out_f32=(T != torch.bfloat16)).to(T))
            h = h + y.view(n, t, d)
        return h[:, -1] if last_only else h[:, self.clip_length:]

    def _last_token_layer(self, lyr, h, T):
        """One TransformerLayer (clipcap.py:39-100) evaluated for the last token only.  h [n, t, d] f32 -> [n, d] f32."""
        n, t, d = h.shape
        a = lyr.attn
        H, dh = a.num_heads, d // a.num_heads
        y = layers.layer_norm(h.reshape(n * t, d), lyr.norm1.weight, lyr.norm1.bias, T)           # all tokens: keys / values
        if T == torch.bfloat16 and dh % 8 == 0 and dh <= 128 and t <= 128 and a.to_keys_values.bias is None:
            # throughput path: both projections emit bf16, the one-query attention is one kernel per direction
            kv = a.to_keys_values(y, out_f32=False)                                                # [n*t, 2d]: keys | values
            q = a.to_queries(y.view(n, t, d)[:, -1].contiguous(), out_f32=False)                   # the one query row [n, d]
            o = layers.last_token_attention(q, kv, t, H, a.scale)
            hl = a.project(o, residual=h[:, -1].contiguous())                                      # (the f32 residual add rides in the epilogue)
            y = layers.layer_norm(hl, lyr.norm2.weight, lyr.norm2.bias, T)
            return lyr.mlp(y, hl)
        kv = a.to_keys_values(y).view(n, t, 2, H, dh)
        q = a.to_queries(y.view(n, t, d)[:, -1].contiguous()).view(n, H, 1, dh)                   # the one query row
        k, v = kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)                   # [n, H, t, dh] f32
        att = torch.softmax((q * k).sum(-1) * a.scale, dim=-1)                                    # [n, H, t]
        o = (att.unsqueeze(-1) * v).sum(2).reshape(n, d)
        hl = h[:, -1] + a.project(o.to(T))
        y = layers.layer_norm(hl.contiguous(), lyr.norm2.weight, lyr.norm2.bias, T)
        return hl + lyr.mlp.fc2(lyr.mlp.fc1(y, relu=True).to(T))


def v2l(prefix, model):
    """clipcap.py:714-719: the LAST of the 40 mapped tokens."""
    if isinstance(model, TransformerMapper):
        return model(prefix, last_only=True)
    prefix_length, size = 40, 768
    embed = model(prefix).reshape(-1, prefix_length, size)[:, -1, :]
    return embed.reshape(embed.shape[0], -1)
