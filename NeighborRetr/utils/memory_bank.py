"""`NeighborRetr.utils.memory_bank` of the reference (memory_bank.py:22-260) -> neighborretr_amd.training."""
from neighborretr_amd.training import MemoryBankManager  # noqa: F401
