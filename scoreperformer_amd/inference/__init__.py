from .generators import PerformanceData, ScorePerformerGenerator

__all__ = ["PerformanceData", "ScorePerformerGenerator"]
