from neighborretr_amd.modeling import NeighborRetr  # noqa: F401
