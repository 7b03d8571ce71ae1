from .ahc import AHC

__all__ = ["AHC"]
