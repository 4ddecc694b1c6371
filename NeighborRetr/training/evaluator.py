"""`NeighborRetr.training.evaluator` of the reference (evaluator.py:66-291) -> neighborretr_amd.training."""
from neighborretr_amd.training import eval_epoch  # noqa: F401
