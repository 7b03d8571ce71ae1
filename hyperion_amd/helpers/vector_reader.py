"""Loads the vectors named by a key list (hyperion/helpers/vector_reader.py)."""
from ..io import RandomAccessDataReaderFactory as DRF
from ..utils import SCPList


class VectorReader(object):
    def __init__(self, v_file, key_file, preproc=None, vlist_sep=" "):
        self.r = DRF.create(v_file)
        self.scp = SCPList.load(key_file, sep=vlist_sep)
        self.preproc = preproc

    def read(self):
        """One [n, d] array when all records are vectors of one size; otherwise a list of per-record
        arrays, each transformed on its own (vector_reader.py:28-41)."""
        try:
            x = self.r.read(self.scp.key, squeeze=True)
            return x if self.preproc is None else self.preproc.predict(x)
        except AssertionError:
            x = self.r.read(self.scp.key, squeeze=False)
            if self.preproc is not None:
                x = [self.preproc.predict(v[None, :] if v.ndim == 1 else v) for v in x]
            return x

    def read_device(self):
        return self.r.read_device(self.scp.key)

    @staticmethod
    def filter_args(**kwargs):
        return dict((k, kwargs[k]) for k in ("vlist_sep",) if k in kwargs)
    @staticmethod
    def add_class_args(parser, prefix=None):
        """hyperion/helpers/vector_reader.py:48-60.  The reference nests the option in a
        jsonargparse sub-parser when a prefix is given; with plain argparse the same option is spelled
        `--<prefix>.vlist-sep`."""
        p1 = "--" if prefix is None else "--" + prefix + "."
        parser.add_argument(p1 + "vlist-sep", default=" ", help="utterance file field separator")

    add_argparse_args = add_class_args
