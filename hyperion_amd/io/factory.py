"""Read specifiers `scp[,options]:file` -> reader objects (hyperion/io/rw_specifiers.py:323-420,
hyperion/io/data_rw_factory.py:139-173): ark archives behind scp files, HDF5 files, and scp files
over HDF5 files (the HDF5 side through the package's own reader); write specifiers
`ark|h5[,scp][,f]:archive[,script]` -> writer objects (rw_specifiers.py:84-224,
data_rw_factory.py:23-61)."""
from .ark import ArkDataWriter, RandomAccessArkDataReader
from .h5 import H5DataWriter, RandomAccessH5FileDataReader, RandomAccessH5ScriptDataReader
from .rw_specifiers import ArchiveType, RSpecifier, RSpecType, WSpecifier


class DataWriterFactory(object):
    @staticmethod
    def parse(wspecifier):
        """-> (kind 'ark' or 'h5', archive path, script path or None, flush)."""
        w = wspecifier if isinstance(wspecifier, WSpecifier) else WSpecifier.create(wspecifier)
        if w.archive is None or w.archive_type not in (ArchiveType.H5, ArchiveType.ARK):
            raise ValueError("wspecifier %s names no ark or h5 archive" % (wspecifier,))
        if not w.binary:
            raise NotImplementedError("text archives are not written")
        return ("h5" if w.archive_type == ArchiveType.H5 else "ark"), w.archive, w.script, w.flush

    @staticmethod
    def create(wspecifier, compress=False, compression_method="auto", scp_sep=" "):
        kind, archive, script, flush = DataWriterFactory.parse(wspecifier)
        cls = H5DataWriter if kind == "h5" else ArkDataWriter
        return cls(archive, script, flush=flush, compress=compress, compression_method=compression_method,
                   scp_sep=scp_sep)

    @staticmethod
    def filter_args(**kwargs):
        return dict((k, kwargs[k]) for k in ("scp_sep", "compress", "compression_method") if k in kwargs)


    compression_methods = ["auto", "speech-feat", "2byte-auto", "2byte-signed-integer", "1byte-auto",
                           "1byte-unsigned-integer", "1byte-0-1", "speech-feat-t"]

    @staticmethod
    def add_class_args(parser, prefix=None):
        """Writer options of the scripts (hyperion/io/data_rw_factory.py:63-77; the choices are those of
        hyperion/utils/kaldi_matrix.py:244-253).  The reference nests them in a jsonargparse sub-parser
        when a prefix is given; with plain argparse they are spelled `--<prefix>.<option>`."""
        p1 = "--" if prefix is None else "--" + prefix + "."
        parser.add_argument(p1 + "scp-sep", default=" ", help="scp file field separator")
        parser.add_argument(p1 + "compress", default=False, action="store_true")
        parser.add_argument(p1 + "compression-method", default="auto", choices=DataWriterFactory.compression_methods)

    add_argparse_args = add_class_args


class RandomAccessDataReaderFactory(object):
    @staticmethod
    def parse(rspecifier):
        """-> (kind, path, permissive) with kind 'scp' (Ark data), 'scp-h5', 'ark' or 'h5'."""
        r = rspecifier if isinstance(rspecifier, RSpecifier) else RSpecifier.create(rspecifier)
        if r.archive_type not in (ArchiveType.H5, ArchiveType.ARK):
            raise ValueError("rspecifier %s: only ark and h5 data is read here" % (rspecifier,))
        h5 = r.archive_type == ArchiveType.H5
        if r.spec_type == RSpecType.SCRIPT:
            return ("scp-h5" if h5 else "scp"), r.archive, r.permissive
        return ("h5" if h5 else "ark"), r.archive, r.permissive

    @staticmethod
    def create(rspecifier, path_prefix=None, transform=None, scp_sep=" "):
        kind, path, permissive = RandomAccessDataReaderFactory.parse(rspecifier)
        if kind == "scp":
            return RandomAccessArkDataReader(path, path_prefix, transform=transform, permissive=permissive,
                                             scp_sep=scp_sep)
        if kind == "ark":
            raise ValueError("Random access to Ark file %s needs a script file" % path)
        if kind == "scp-h5":
            return RandomAccessH5ScriptDataReader(path, path_prefix, scp_sep=scp_sep, transform=transform,
                                                  permissive=permissive)
        return RandomAccessH5FileDataReader(path, transform=transform, permissive=permissive)

    @staticmethod
    def filter_args(**kwargs):
        return dict((k, kwargs[k]) for k in ("scp_sep", "path_prefix") if k in kwargs)

    @staticmethod
    def add_class_args(parser, prefix=None):
        """Reader options of the scripts (data_rw_factory.py:180-198); an option another helper has
        already given the parser is left as it is, as there."""
        import argparse

        p1 = "--" if prefix is None else "--" + prefix + "."
        try:
            parser.add_argument(p1 + "scp-sep", default=" ", help="scp file field separator")
        except argparse.ArgumentError:
            pass
        parser.add_argument(p1 + "path-prefix", default=None, help="scp file_path prefix")

    add_argparse_args = add_class_args
