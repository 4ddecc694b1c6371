"""`NeighborRetr.training.trainer` of the reference (trainer.py:18-221) -> neighborretr_amd.training."""
from neighborretr_amd.training import train_epoch  # noqa: F401
