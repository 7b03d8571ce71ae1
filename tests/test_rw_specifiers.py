"""Read / write specifier objects: the cases of the reference's own test
(tests/hyperion/io/test_rw_specifiers.py) as a table, plus the error paths and the factories that
parse through the same classes."""
import pytest

from hyperion_amd.io import (ArchiveType, DataWriterFactory, RandomAccessDataReaderFactory, RSpecifier, RSpecType,
                             WSpecifier, WSpecType)

A, R, W = ArchiveType, RSpecType, WSpecType

WCASES = [
    ("file.h5", (W.ARCHIVE, "file.h5", None, A.H5)),
    ("h5:file.h5", (W.ARCHIVE, "file.h5", None, A.H5)),
    ("ark:file.ark", (W.ARCHIVE, "file.ark", None, A.ARK)),
    ("ark,b,nf:file.ark", (W.ARCHIVE, "file.ark", None, A.ARK)),
    ("ark,t,f,p:file.ark", (W.ARCHIVE, "file.ark", None, A.ARK, False, True, True)),
    ("scp:file.scp", (W.SCRIPT, None, "file.scp", None)),
    ("ark,t,scp:file.ark,file.scp", (W.BOTH, "file.ark", "file.scp", A.ARK, False)),
    ("h5,scp:file.h5,file.scp", (W.BOTH, "file.h5", "file.scp", A.H5)),
    ("audio,scp:file.flac,file.scp", (W.BOTH, "file.flac", "file.scp", A.AUDIO)),
    ("h5,t,p:file.h5", (W.ARCHIVE, "file.h5", None, A.H5)),  # t and p only mean something for Ark
]


@pytest.mark.parametrize("spec,args", WCASES)
def test_wspecifier(spec, args):
    got = WSpecifier.create(spec)
    assert got == WSpecifier(*args) and not (got != WSpecifier(*args)) and got.__cmp__(WSpecifier(*args)) == 0


def test_rspecifier(tmp_path):
    assert RSpecifier.create("file.h5") == RSpecifier(R.ARCHIVE, "file.h5", A.H5)
    assert RSpecifier.create("h5:file.h5") == RSpecifier(R.ARCHIVE, "file.h5", A.H5)
    assert RSpecifier.create("ark:file.ark") == RSpecifier(R.ARCHIVE, "file.ark", A.ARK)
    assert RSpecifier.create("ark,o,s,cs,p,bg:file.ark") == RSpecifier(R.ARCHIVE, "file.ark", A.ARK, True, True, True,
                                                                      True, True)
    assert RSpecifier.create("h5,o,s,p:file.h5") == RSpecifier(R.ARCHIVE, "file.h5", A.H5, permissive=True)
    assert RSpecifier.create("ark:file.ark") != RSpecifier.create("ark,p:file.ark")
    scp = str(tmp_path / "file.scp")
    for line, kind in (("key file1.ark:0\n", A.ARK), ("key file1.h5\n", A.H5), ("key file1.h5[3:9]\n", A.H5),
                       ("key file1.flac:0[0:10]\n", A.AUDIO), ("key some_pipe_or_other\n", A.ARK)):
        with open(scp, "w") as f:
            f.write(line)
        got = RSpecifier.create("scp:" + scp)
        assert got == RSpecifier(R.SCRIPT, scp, kind), line
        assert got.script == scp


@pytest.mark.parametrize("make,spec,err", [
    (WSpecifier.create, "ark,zz:file.ark", ValueError), (WSpecifier.create, "b:file.ark", ValueError),
    (WSpecifier.create, "a:b:c", ValueError), (WSpecifier.create, "ark,h5:a,b", AssertionError),
    (WSpecifier.create, "ark,scp:a", AssertionError), (WSpecifier.create, "scp,scp:a,b", AssertionError),
    (WSpecifier.create, "", AssertionError),
    (RSpecifier.create, "ark,zz:file.ark", ValueError), (RSpecifier.create, "p:file.ark", AssertionError),
    (RSpecifier.create, "ark,scp:file.ark", AssertionError), (RSpecifier.create, "a:b:c", ValueError),
])
def test_malformed_specifiers(make, spec, err):
    with pytest.raises(err):
        make(spec)


def test_factories_parse_through_the_specifier_classes(tmp_path):
    assert DataWriterFactory.parse("ark,scp,f:a.ark,a.scp") == ("ark", "a.ark", "a.scp", True)
    assert DataWriterFactory.parse("x.h5") == ("h5", "x.h5", None, False)
    assert DataWriterFactory.parse(WSpecifier.create("h5,scp:a.h5,a.scp")) == ("h5", "a.h5", "a.scp", False)
    with pytest.raises(NotImplementedError):
        DataWriterFactory.parse("ark,t:a.ark")
    for bad in ("scp:a.scp", "audio:a.flac"):
        with pytest.raises(ValueError):
            DataWriterFactory.parse(bad)
    scp = str(tmp_path / "v.scp")
    with open(scp, "w") as f:
        f.write("u1 v.ark:5\n")
    assert RandomAccessDataReaderFactory.parse("scp,p:" + scp) == ("scp", scp, True)
    with open(scp, "w") as f:
        f.write("u1 v.h5\n")
    assert RandomAccessDataReaderFactory.parse("scp:" + scp) == ("scp-h5", scp, False)
    assert RandomAccessDataReaderFactory.parse("v.h5") == ("h5", "v.h5", False)
    assert RandomAccessDataReaderFactory.parse("ark:v.ark") == ("ark", "v.ark", False)
    with pytest.raises(ValueError):
        RandomAccessDataReaderFactory.parse("rttm:v.rttm")
