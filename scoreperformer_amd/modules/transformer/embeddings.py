"""General transformer embeddings: same classes/state_dict keys as `modules/transformer/embeddings.py:11-325`.

The per-key table classes are parameter containers; their tables are built for ALL keys of a TupleTokenEmbeddings in one
HIP launch (`functional.TableBuildFn`, see models/scoreperformer/embeddings.py).  `.weight` builds a single table on
demand with the same kernel (stand-alone use).
"""
import math
from typing import Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from ... import functional as F_
from ...utils.amp import no_autocast


class DiscreteContinuousEmbedding(nn.Module):
    dense = False

    def __init__(self, num_embeddings: int, embedding_dim: int, discrete: bool = True, continuous: bool = True,
                 discrete_ids: Optional[Union[list, Tensor]] = None, token_values: Optional[Union[list, Tensor]] = None,
                 padding_idx: Optional[int] = None, activation=None, _weight: Optional[Tensor] = None, device=None, dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        if discrete_ids is not None:
            if not isinstance(discrete_ids, Tensor):
                discrete_ids = torch.tensor(list(discrete_ids))
            discrete_ids = discrete_ids.reshape(-1).to(device=device, dtype=torch.long)
        self.discrete_ids = discrete_ids
        if padding_idx is not None:
            if padding_idx > 0:
                assert padding_idx < num_embeddings, 'Padding_idx must be within num_embeddings'
            elif padding_idx < 0:
                assert padding_idx >= -num_embeddings, 'Padding_idx must be within num_embeddings'
                padding_idx = num_embeddings + padding_idx
        self.padding_idx = padding_idx
        assert discrete or continuous, '`DiscreteContinuousEmbedding` should be at least discrete or continuous'
        self.discrete, self.continuous = discrete, continuous

        self.index_weight = None
        if self.has_discrete:
            if _weight is None:
                self.index_weight = nn.Parameter(torch.empty((num_embeddings, embedding_dim), **fk))
            else:
                assert list(_weight.shape) == [num_embeddings, embedding_dim]
                self.index_weight = nn.Parameter(_weight)
        self.value_layer = None
        self.activation = None
        if self.continuous:
            if token_values is not None:
                if not isinstance(token_values, Tensor):
                    token_values = torch.tensor(list(token_values), dtype=torch.float32)
            else:
                token_values = torch.linspace(0., 1., num_embeddings)
            token_values = token_values.reshape(-1, 1).to(**fk)
            self.value_layer = nn.Linear(1, embedding_dim, bias=False, **fk)
            self.activation = activation
            if activation is not None:
                raise NotImplementedError("custom activation on DiscreteContinuousEmbedding is not used by any recipe")
        self.register_buffer('token_values', token_values)
        self._value_weight = None
        if _weight is None:
            self.reset_parameters()

    INIT_STD = 1e-2   # both parameter families start as N(0, 0.01) (modules/transformer/embeddings.py:76-81)

    def reset_parameters(self) -> None:
        for used, name in ((self.has_discrete, "index_weight"), (self.continuous, "value_layer")):
            if used:
                target = getattr(self, name)
                nn.init.normal_(target if isinstance(target, Tensor) else target.weight, std=self.INIT_STD)
        self._fill_padding_idx_with_zero()

    @torch.no_grad()
    def _fill_padding_idx_with_zero(self) -> None:
        """The padding id embeds to zero: its index row and its token value are cleared (embeddings.py:83-90)."""
        pad = self.padding_idx
        if pad is None:
            return
        if self.has_discrete:
            self.index_weight[pad].zero_()
        if self.continuous and self.token_values is not None:
            self.token_values[pad].zero_()

    @property
    def has_discrete(self):
        return self.discrete or self.discrete_ids is not None

    @property
    def ids_mask(self) -> int:
        m = 0
        if self.discrete_ids is not None:
            for i in self.discrete_ids.tolist():
                if i >= 32:
                    raise NotImplementedError("discrete_ids >= 32 are not supported by the table kernel")
                m |= 1 << i
        return m

    def table_params(self):
        """(tv, w0, b0, w1, b1, iw) in the order the table-build kernel expects."""
        if self.dense:
            l0, l1 = self.value_layer[0][0], self.value_layer[1][0]
            return self.token_values, l0.weight, l0.bias, l1.weight, l1.bias, self.index_weight
        return self.token_values, self.value_layer.weight, None, None, None, self.index_weight

    @property
    def weight(self):
        if not self.continuous:
            return self.index_weight
        from ...models.scoreperformer.embeddings import build_tables
        return build_tables([self])[0]

    @no_autocast
    def forward(self, tokens: Optional[Tensor] = None, values: Optional[Tensor] = None) -> Tensor:
        if values is not None:
            raise NotImplementedError("explicit `values` are not used on the ScorePerformer hot path")
        return F_.EmbedFn.apply(tokens[..., None] if tokens.ndim == 2 else tokens.reshape(1, -1, 1), None, None, 1e-5,
                                self.weight).float().view(*tokens.shape, self.embedding_dim)

    def extra_repr(self) -> str:
        s = f'{self.num_embeddings}, {self.embedding_dim}'
        if self.padding_idx is not None:
            s += f', padding_idx={self.padding_idx}'
        return s


class DiscreteDenseContinuousEmbedding(DiscreteContinuousEmbedding):
    dense = True

    def __init__(self, num_embeddings: int, embedding_dim: int, depth: int = 2, discrete: bool = True, continuous: bool = True,
                 discrete_ids=None, token_values=None, padding_idx: Optional[int] = None, _weight: Optional[Tensor] = None,
                 device=None, dtype=None):
        super().__init__(num_embeddings=num_embeddings, embedding_dim=embedding_dim, discrete=discrete, continuous=continuous,
                         discrete_ids=discrete_ids, token_values=token_values, padding_idx=padding_idx, device=device, dtype=dtype)
        if depth != 2:
            raise NotImplementedError("the table kernel implements the depth-2 value MLP used by every recipe")
        if self.continuous:
            fk = {"device": device, "dtype": dtype}
            self.value_layer = nn.Sequential(
                nn.Sequential(nn.Linear(1, embedding_dim, **fk), nn.Mish()),
                nn.Sequential(nn.Linear(embedding_dim, embedding_dim, **fk), nn.Identity()))
            self.reset_parameters()

    def reset_parameters(self) -> None:
        if self.has_discrete:
            nn.init.normal_(self.index_weight, std=1e-2)
        if self.continuous and isinstance(self.value_layer, nn.Sequential):
            for module in self.value_layer.modules():
                if isinstance(module, nn.Linear):
                    nn.init.normal_(module.weight, std=1e-2)
        self._fill_padding_idx_with_zero()


class AbsolutePositionalEmbedding(nn.Module):
    def __init__(self, dim, max_seq_len):
        super().__init__()
        self.dim, self.scale, self.max_seq_len = dim, dim ** -0.5, max_seq_len
        self.emb = nn.Embedding(max_seq_len, dim)

    @no_autocast
    def forward(self, x: Tensor, pos: Optional[Tensor] = None):
        seq_len = x.shape[1]
        assert seq_len <= self.max_seq_len
        if pos is None:
            return self.emb.weight[:seq_len] * self.scale
        return self.emb.weight[pos] * self.scale

    def extra_repr(self) -> str:
        return f'dim={self.dim}'


class FixedPositionalEmbedding(nn.Module):
    """Sinusoidal table [sin(p w_k) | cos(p w_k)], w_k = 10000^(-2k/dim)  (modules/transformer/embeddings.py:245-265)."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        exponents = torch.arange(0, dim, 2, dtype=torch.float32) / dim
        self.register_buffer('inv_freq', torch.reciprocal(torch.pow(10000.0, exponents)))

    @no_autocast
    def forward(self, x: Tensor, pos: Optional[Tensor] = None, seq_dim: int = 1, offset: int = 0):
        positions = torch.arange(x.shape[seq_dim], device=x.device) if pos is None else pos
        angles = torch.outer(positions.to(self.inv_freq.dtype).reshape(-1) + offset, self.inv_freq)
        angles = angles.reshape(*positions.shape, -1)
        return torch.cat((torch.sin(angles), torch.cos(angles)), dim=-1)

    def extra_repr(self) -> str:
        return f'dim={self.dim}'


class ALiBiPositionalBias(nn.Module):
    """Per-head slopes; the (h, i, j) bias tensor is never materialised on the hot path -- the attention kernels compute
    slope_h * -|j - (i + j_len - i_len)| in registers.  `get_bias` / `forward` are kept for API parity."""

    def __init__(self, heads: int, total_heads: int, symmetric: bool = True):
        super().__init__()
        self.heads, self.total_heads, self.symmetric = heads, total_heads, symmetric
        if not symmetric:
            raise NotImplementedError("asymmetric ALiBi is unreachable in the reference (attention.py:85) and unsupported")
        slopes = torch.tensor(self._compute_slopes(heads), dtype=torch.float32).view(-1, 1, 1)
        self.register_buffer('slopes', slopes, persistent=False)

    @staticmethod
    def _compute_slopes(heads):
        def slopes_power_of_2(n):
            start = (2 ** (-2 ** -(math.log2(n) - 3)))
            return [start * start ** i for i in range(n)]

        if math.log2(heads).is_integer():
            return slopes_power_of_2(heads)
        closest = 2 ** math.floor(math.log2(heads))
        return slopes_power_of_2(closest) + slopes_power_of_2(2 * closest)[0::2][:heads - closest]

    def get_bias(self, i: int, j: int, k: int = 0):
        """-|key - query| for queries k .. k+i-1 against keys 0 .. j-1, shape [1, i, j] (embeddings.py:294-297)."""
        dev = self.slopes.device
        query = torch.arange(k, k + i, dtype=torch.int, device=dev).view(1, i, 1)
        key = torch.arange(j, dtype=torch.int, device=dev).view(1, 1, j)
        return (key - query).abs().neg()

    def get_slopes(self):
        return self.slopes

    def padded_slopes(self):
        """[total_heads] slopes (zero for heads beyond `heads`, embeddings.py:307-308)."""
        s = self.get_slopes().reshape(-1)
        if self.total_heads - s.shape[0] > 0:
            s = F.pad(s, (0, self.total_heads - s.shape[0]))
        return s

    @no_autocast
    def forward(self, i: int, j: int, k: int = 0, bias: Optional[Tensor] = None):
        if bias is not None and bias.shape[-2] >= i and bias.shape[-1] >= j - k:
            bias = bias[..., :i, :j]
        else:
            bias = self.get_bias(i, j, k)
        return self.padded_slopes().view(-1, 1, 1) * bias


class LearnedALiBiPositionalBias(ALiBiPositionalBias):
    def __init__(self, heads: int, total_heads: int, symmetric: bool = True):
        super().__init__(heads, total_heads, symmetric)
        self.learned_logslopes = nn.Parameter(torch.log(self.slopes))

    def get_slopes(self):
        p = self.learned_logslopes
        if p.is_cuda and p.requires_grad and torch.is_grad_enabled():
            from ... import functional as F_
            return F_.ExpSlopesFn.apply(p)
        return p.exp()
