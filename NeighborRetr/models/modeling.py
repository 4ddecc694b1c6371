"""`NeighborRetr.models.modeling` of the reference -> neighborretr_amd.modeling (HIP-backed)."""
from neighborretr_amd.modeling import *  # noqa: F401,F403
from neighborretr_amd.modeling import NeighborRetr, allgather, default_config  # noqa: F401
from neighborretr_amd.until_module import AllGather, AllGather2  # noqa: F401
