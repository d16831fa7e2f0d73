from .attend import AttentionIntermediates, Attend
from .attention import AttentionSharedIntermediates, AttentionConfig, Attention
from .embeddings import (
    DiscreteContinuousEmbedding, DiscreteDenseContinuousEmbedding, AbsolutePositionalEmbedding,
    FixedPositionalEmbedding, ALiBiPositionalBias, LearnedALiBiPositionalBias
)
from .feedforward import FeedForwardConfig, FeedForward
from .transformer import (
    TransformerIntermediates, TransformerRegistry, TransformerConfig, Transformer,
    EncoderConfig, Encoder, DecoderConfig, Decoder
)

DEFAULT_DIM_HEAD = 64
