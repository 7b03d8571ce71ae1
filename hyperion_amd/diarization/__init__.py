from .diar_ahc_plda import DiarAHCPLDA

__all__ = ["DiarAHCPLDA"]
