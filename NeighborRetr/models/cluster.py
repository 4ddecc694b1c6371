"""`NeighborRetr.models.cluster` of the reference -> neighborretr_amd.cluster."""
from neighborretr_amd.cluster import CTM, TCBlock, TCAttention, TokenConv, dpc_knn_assign, merge_by_cluster  # noqa: F401
