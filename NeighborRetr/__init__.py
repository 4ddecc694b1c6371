"""Import-path shim: the reference's package layout (`NeighborRetr.models.modeling`, ...) mapped
onto the MI355X-native implementation in `neighborretr_amd`, so code written against the reference
(`from NeighborRetr.models.modeling import NeighborRetr, AllGather`) runs unchanged."""
