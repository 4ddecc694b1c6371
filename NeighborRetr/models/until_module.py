"""`NeighborRetr.models.until_module` of the reference -> neighborretr_amd.until_module."""
from neighborretr_amd.until_module import (AllGather, AllGather2, CentralityWeightingLoss, KLDivergenceLoss,  # noqa: F401
                                           NeighborAdjustingLoss, UniformRegularizationLoss)
