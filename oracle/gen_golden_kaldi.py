"""TEST INFRASTRUCTURE (in-container only).  Generates tests/golden/kaldi_compressed/: binary Kaldi archives of compressed matrices written by the
reference's own writer (hyperion/io/ark_data_writer.py with compress=True, every compression method of
hyperion/utils/kaldi_matrix.py:255-263) and what the reference's reader returns for them (whole
records and row ranges).  Run in the build container, where /root/reference is importable:

    python -B oracle/gen_golden_kaldi.py

The archives and expected.npz are data; nothing of the reference is copied."""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HERE = os.path.join(ROOT, "tests", "golden", "kaldi_compressed")
METHODS = ["auto", "speech-feat", "speech-feat-t", "2byte-auto", "2byte-signed-integer", "1byte-auto",
           "1byte-unsigned-integer", "1byte-0-1"]


def matrices():
    rng = np.random.RandomState(17)
    m = {
        "feat40x13": (rng.randn(40, 13) * np.linspace(0.5, 20, 13) + np.linspace(-30, 30, 13)).astype(np.float32),
        "tall12x7": (rng.randn(12, 7) * 3).astype(np.float32),
        "rows4": rng.randn(4, 5).astype(np.float32),
        "rows3": rng.randn(3, 5).astype(np.float32),
        "rows2": rng.randn(2, 6).astype(np.float32),
        "rows1": rng.randn(1, 6).astype(np.float32),
        "const9x4": np.full((9, 4), 2.5, dtype=np.float32),
        "ints10x6": rng.randint(0, 256, size=(10, 6)).astype(np.float32),
        "unit11x5": rng.rand(11, 5).astype(np.float32),
    }
    m["feat40x13"][5, 3] = 300.0  # one far outlier in a column
    return m


def main():
    from oracle.gen_golden_io import load_reference_io

    load_reference_io()
    from hyperion.io.ark_data_reader import RandomAccessArkDataReader
    from hyperion.io.ark_data_writer import ArkDataWriter

    os.makedirs(HERE, exist_ok=True)
    mats = matrices()
    expected = {}
    for method in METHODS:
        ark = os.path.join(HERE, "%s.ark" % method)
        scp = os.path.join(HERE, "%s.scp" % method)
        cwd = os.getcwd()
        os.chdir(HERE)  # the script stores the archive path as given: keep it relative
        try:
            w = ArkDataWriter("./%s.ark" % method, "./%s.scp" % method, compress=True, compression_method=method)
            keys = sorted(mats)
            w.write(keys, [mats[k] for k in keys])
            w.close()
            r = RandomAccessArkDataReader("./%s.scp" % method)
            for k in keys:
                expected["%s/%s" % (method, k)] = r.read([k])[0]
            expected["%s/feat40x13/rows7+9" % method] = r.read(["feat40x13"], row_offset=7, num_rows=9)[0]
            expected["%s/tall12x7/rows3+0" % method] = r.read(["tall12x7"], row_offset=3)[0]
            r.close()
        finally:
            os.chdir(cwd)
        assert os.path.getsize(ark) > 0 and os.path.getsize(scp) > 0
    np.savez_compressed(os.path.join(HERE, "expected.npz"), **expected)
    np.savez_compressed(os.path.join(HERE, "inputs.npz"), **mats)
    print("wrote", len(expected), "expected arrays for", len(METHODS), "methods")


if __name__ == "__main__":
    main()
