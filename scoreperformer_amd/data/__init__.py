from .collators import (SeqInputs, SeqSegments, SegmentBounds, MixedLMScorePerformanceInputs, MixedLMScorePerformanceCollator)

__all__ = ["SeqInputs", "SeqSegments", "SegmentBounds", "MixedLMScorePerformanceInputs", "MixedLMScorePerformanceCollator"]
