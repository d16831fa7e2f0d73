"""Operator API mirror of `scoreperformer.modules` (same names, constructor/forward contracts), HIP-backed."""
from .constructor import Constructor, ModuleConfig, VariableModuleConfig, Registry, merge
from .layers import Residual, AdaptiveLayerNorm, LayerNorm
from .sampling import top_p, top_k, top_a, filter_logits_and_sample
