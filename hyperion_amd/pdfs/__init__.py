from .plda import PLDABase, SPLDA, FRPLDA, PLDA

__all__ = ["PLDABase", "SPLDA", "FRPLDA", "PLDA"]
