"""Kaldi-style read / write specifiers as objects, with the reference's class, member and attribute
names (hyperion/io/rw_specifiers.py:12-20, 75-248, 291-436).

    wspecifier   [kind][,scp][,t|b][,f|nf][,p]:archive[,script]     kind in h5 ark audio segments rttm
    rspecifier   kind|scp[,o][,s][,cs][,p][,bg]:file

A bare path is an HDF5 archive.  For `scp:` read specifiers the archive type is taken from the second
field of the script's first line.  The factories in hyperion_amd/io/factory.py parse through these
classes; they build readers and writers for the `h5` and `ark` types only.
"""
import re
from enum import Enum


class ArchiveType(Enum):
    H5 = 0
    ARK = 1
    AUDIO = 2
    SEGMENT_LIST = 3
    RTTM = 4


class WSpecType(Enum):
    NO = 0
    ARCHIVE = 1
    SCRIPT = 2
    BOTH = 3


class RSpecType(Enum):
    NO = 0
    ARCHIVE = 1
    SCRIPT = 2


_KINDS = {"h5": ArchiveType.H5, "ark": ArchiveType.ARK, "audio": ArchiveType.AUDIO,
          "segments": ArchiveType.SEGMENT_LIST, "rttm": ArchiveType.RTTM}


def _split(spec):
    """'options:paths' -> (option list or None for a bare path, paths string)."""
    fields = spec.strip().split(":")
    if len(fields) == 1:
        assert len(fields[0]) > 0
        return None, fields[0]
    if len(fields) != 2:
        raise ValueError("Two many fields (%d>2) in specifier %s" % (len(fields), spec))
    return fields[0].strip().split(","), fields[1].strip()


class _Spec(object):
    _compared = ()

    def __eq__(self, other):
        return all(getattr(self, k) == getattr(other, k) for k in self._compared)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __cmp__(self, other):
        return 0 if self.__eq__(other) else 1

    __hash__ = None

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join("%s=%r" % (k, getattr(self, k)) for k in self._compared))


class WSpecifier(_Spec):
    """Where and how a keyed set of arrays is written: `archive` (data file) and / or `script` (index
    file), `archive_type`, and for Ark archives `binary` (False with option t) and `permissive`;
    `flush` after every record with option f."""

    _compared = ("archive", "script", "spec_type", "archive_type", "binary", "flush", "permissive")

    def __init__(self, spec_type, archive, script, archive_type=ArchiveType.H5, binary=True, flush=False,
                 permissive=False):
        self.spec_type = spec_type
        self.archive = archive
        self.script = script
        self.archive_type = archive_type
        self.binary = binary
        self.flush = flush
        self.permissive = permissive

    @classmethod
    def create(cls, wspecifier):
        options, paths = _split(wspecifier)
        if options is None:
            return cls(WSpecType.ARCHIVE, paths, None)
        paths = paths.split(",")
        archive = script = archive_type = None
        flush = False
        taken = 0
        for option in options:
            if option in _KINDS or option == "scp":
                assert len(paths) > taken, "wspecifier %s names fewer files than kinds" % wspecifier
                if option == "scp":
                    assert script is None, "Repeated scp in wspecifier %s" % wspecifier
                    script = paths[taken]
                else:
                    assert archive_type is None and archive is None, "Repeated archive kind in wspecifier %s" % wspecifier
                    archive_type, archive = _KINDS[option], paths[taken]
                taken += 1
            elif option == "f":
                flush = True
            elif option not in ("b", "t", "nf", "p"):
                raise ValueError("Wrong wspecifier options %s" % ",".join(options))
        if archive is None and script is None:
            raise ValueError("Wrong wspecifier %s " % wspecifier)
        spec_type = (WSpecType.SCRIPT if archive is None else
                     WSpecType.ARCHIVE if script is None else WSpecType.BOTH)
        is_ark = archive_type == ArchiveType.ARK  # text mode and permissive only mean something for Ark
        return cls(spec_type, archive, script, archive_type, binary=not (is_ark and "t" in options), flush=flush,
                   permissive=is_ark and "p" in options)


class RSpecifier(_Spec):
    """What a keyed set of arrays is read from: `archive` (the data file, or the script file for
    spec_type SCRIPT), `archive_type`, `permissive` (option p), and for Ark data the stream hints
    `once` (o), `is_sorted` (s), `called_sorted` (cs), `background` (bg)."""

    _compared = ("spec_type", "archive", "archive_type", "once", "is_sorted", "called_sorted", "permissive",
                 "background")

    def __init__(self, spec_type, archive, archive_type=ArchiveType.H5, once=False, is_sorted=False,
                 called_sorted=False, permissive=False, background=False):
        self.spec_type = spec_type
        self.archive = archive
        self.archive_type = archive_type
        self.once = once
        self.is_sorted = is_sorted
        self.called_sorted = called_sorted
        self.permissive = permissive
        self.background = background

    @property
    def script(self):
        return self.archive

    @staticmethod
    def _type_of_script(path):
        """Archive type from the second field of the script's first line: `x.h5` (optionally with a
        row range) is HDF5, `x.ark:offset` Ark, an audio file with a range audio, anything else Ark."""
        with open(path, "r") as f:
            second = f.readline().strip().split(" ")[1]
        if re.match(r".*\.h5(?:.[0-9]+:[0-9]+.)?$", second) is not None:
            return ArchiveType.H5
        if re.match(r".*\.ark:.*$", second) is not None:
            return ArchiveType.ARK
        if re.match(r".*[cvg]:[0-9]+.[0-9]+:[0-9]+.$", second) is not None:
            return ArchiveType.AUDIO
        return ArchiveType.ARK

    @classmethod
    def create(cls, rspecifier):
        options, path = _split(rspecifier)
        if options is None:
            return cls(RSpecType.ARCHIVE, path)
        path = path.split(",")[0]
        spec_type = archive_type = None
        for option in options:
            if option in _KINDS or option == "scp":
                assert spec_type is None, "Repeated kind in rspecifier %s" % rspecifier
                spec_type = RSpecType.SCRIPT if option == "scp" else RSpecType.ARCHIVE
                archive_type = _KINDS.get(option)
            elif option not in ("o", "s", "cs", "p", "bg"):
                raise ValueError("Wrong rspecifier options %s" % ",".join(options))
        assert spec_type is not None, "Wrong rspecifier options %s" % ",".join(options)
        if spec_type == RSpecType.SCRIPT:
            archive_type = cls._type_of_script(path)
        is_ark = archive_type == ArchiveType.ARK
        return cls(spec_type, path, archive_type, once=is_ark and "o" in options, is_sorted=is_ark and "s" in options,
                   called_sorted=is_ark and "cs" in options, permissive="p" in options,
                   background=is_ark and "bg" in options)
