from .embeddings import TupleTokenEmbeddings, TupleTokenLMHead, TupleTokenTiedLMHead
from .evaluator import ScorePerformerEvaluator
from .model import PerformerConfig, Performer, ScorePerformerConfig, ScorePerformer
from .transformer import TupleTransformerConfig, TupleTransformer, TupleTransformerCaches
from .wrappers import ScorePerformerMLMWrapper
