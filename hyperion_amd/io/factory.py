"""Read specifiers `scp[,options]:file` -> reader objects (hyperion/io/rw_specifiers.py:323-420,
hyperion/io/data_rw_factory.py:139-173): ark archives behind scp files, HDF5 files, and scp files
over HDF5 files (the HDF5 side through the package's own reader); write specifiers
`ark|h5[,scp][,f]:archive[,script]` -> writer objects (rw_specifiers.py:84-224,
data_rw_factory.py:23-61)."""
import re

from .ark import ArkDataWriter, RandomAccessArkDataReader
from .h5 import H5DataWriter, RandomAccessH5FileDataReader, RandomAccessH5ScriptDataReader


class DataWriterFactory(object):
    @staticmethod
    def parse(wspecifier):
        """-> (kind 'ark' or 'h5', archive path, script path or None, flush)."""
        fields = wspecifier.strip().split(":")
        if len(fields) == 1:
            assert len(fields[0]) > 0
            return "h5", fields[0], None, False  # a bare path is an HDF5 archive
        if len(fields) != 2:
            raise ValueError("Two many fields (%d>2) in wspecifier %s" % (len(fields), wspecifier))
        paths = fields[1].strip().split(",")
        kind, archive, script, flush, cur = None, None, None, False, 0
        for option in fields[0].strip().split(","):
            if option in ("h5", "ark"):
                assert kind is None, "Repeated h5, ark in wspecifier %s" % wspecifier
                assert len(paths) > cur
                kind, archive = option, paths[cur]
                cur += 1
            elif option == "scp":
                assert script is None, "Repeated scp in wspecifier %s" % wspecifier
                assert len(paths) > cur
                script = paths[cur]
                cur += 1
            elif option == "f":
                flush = True
            elif option == "t":
                raise NotImplementedError("text archives are not written")
            elif option not in ("b", "nf", "p"):
                raise ValueError("Wrong wspecifier options %s" % fields[0])
        if kind is None:
            raise ValueError("wspecifier %s names no ark or h5 archive" % wspecifier)
        return kind, archive, script, flush

    @staticmethod
    def create(wspecifier, compress=False, compression_method="auto", scp_sep=" "):
        kind, archive, script, flush = DataWriterFactory.parse(wspecifier)
        cls = H5DataWriter if kind == "h5" else ArkDataWriter
        return cls(archive, script, flush=flush, compress=compress, compression_method=compression_method,
                   scp_sep=scp_sep)

    @staticmethod
    def filter_args(**kwargs):
        return dict((k, kwargs[k]) for k in ("scp_sep", "compress", "compression_method") if k in kwargs)


class RandomAccessDataReaderFactory(object):
    @staticmethod
    def parse(rspecifier):
        """-> (kind, path, permissive) with kind 'scp' or 'ark' or 'h5'."""
        fields = rspecifier.strip().split(":")
        if len(fields) == 1:
            assert len(fields[0]) > 0
            return "h5", fields[0], False  # a bare path is an HDF5 archive (rw_specifiers.py:326-328)
        if len(fields) != 2:
            raise ValueError("Two many fields (%d>2) in wspecifier %s" % (len(fields), rspecifier))
        kind, permissive = None, False
        for option in fields[0].strip().split(","):
            if option in ("scp", "ark", "h5"):
                assert kind is None
                kind = option
            elif option == "p":
                permissive = True
            elif option not in ("o", "s", "cs", "bg"):
                raise ValueError("Wrong wspecifier options %s" % fields[0])
        assert kind is not None, "Wrong wspecifier options %s" % fields[0]
        path = fields[1].strip()
        if kind == "scp":
            with open(path, "r") as f:
                second = f.readline().strip().split(" ")[1]
            if re.match(r".*\.h5(?:.[0-9]+:[0-9]+.)?$", second) is not None:
                kind = "scp-h5"
        return kind, path, permissive

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
