"""Readers and writers of the vector files the scoring path starts from (subset of hyperion/io)."""
from .ark import ArkDataWriter, RandomAccessArkDataReader  # noqa: F401
from .factory import DataWriterFactory, RandomAccessDataReaderFactory  # noqa: F401
from .h5 import H5DataWriter  # noqa: F401
from .rw_specifiers import ArchiveType, RSpecifier, RSpecType, WSpecifier, WSpecType  # noqa: F401
