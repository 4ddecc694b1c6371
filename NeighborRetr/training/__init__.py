"""`NeighborRetr.training` of the reference -> neighborretr_amd.training."""
