"""`NeighborRetr.utils.comm.is_main_process` of the reference -> neighborretr_amd.training."""
from neighborretr_amd.training import is_main_process  # noqa: F401
