"""Model API mirror of `scoreperformer.models` (`models/__init__.py:1-9`)."""
from .base import Model
from .scoreperformer import Performer, ScorePerformer, ScorePerformerEvaluator

MODELS = {"Performer": Performer, "ScorePerformer": ScorePerformer}
EVALUATORS = {"ScorePerformerEvaluator": ScorePerformerEvaluator}
