"""`NeighborRetr.utils.metrics` of the reference -> neighborretr_amd.metrics."""
from neighborretr_amd.metrics import RetrievalMetrics  # noqa: F401
