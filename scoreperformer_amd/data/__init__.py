from .collators import (SeqInputs, SeqSegments, MixedLMScorePerformanceInputs, MixedLMScorePerformanceCollator)

__all__ = ["SeqInputs", "SeqSegments", "MixedLMScorePerformanceInputs", "MixedLMScorePerformanceCollator"]
