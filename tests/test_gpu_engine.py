"""Host <-> device staging of the numpy API: page-locked buffers of bounded size, arrays larger than a
buffer go in row slices (two buffers in turn), read-only inputs are taken without a warning."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_staged_copies_in_slices(monkeypatch):
    import torch

    from hyperion_amd import engine

    rng = np.random.RandomState(5)
    x = rng.randn(5000, 256)  # float64, 1.28 M elements: the staged path
    monkeypatch.setattr(engine, "STAGE_BYTES", 1 << 20)  # 1 MiB: 1024 fp32 rows / 512 float64 rows per slice
    engine.drop_staging()
    a = engine.upload(x)
    assert a.dtype == torch.float32 and torch.equal(a.cpu(), torch.from_numpy(x).float())
    b = engine.upload64(x)
    assert b.dtype == torch.float64 and torch.equal(b.cpu(), torch.from_numpy(x))
    back = engine.to_host(a, np.float64)
    assert back.dtype == np.float64 and np.array_equal(back, x.astype(np.float32).astype(np.float64))
    view = a[:, :200]  # a column slice of a wider device matrix
    assert np.array_equal(engine.to_host(view, np.float64), back[:, :200])
    for buf in engine._staging.__dict__.values():  # no buffer beyond the cap (or the 4 MiB minimum)
        assert buf.numel() * buf.element_size() <= max(engine.STAGE_BYTES, 8 << 20)
    engine.drop_staging()
    assert not engine._staging.__dict__
    # one slice (the default cap) gives the same bits
    monkeypatch.undo()
    assert torch.equal(engine.upload(x), a) and torch.equal(engine.upload64(x), b)
    assert np.array_equal(engine.to_host(a, np.float64), back)


def test_read_only_input_draws_no_warning():
    import torch

    from hyperion_amd import engine

    x = np.random.RandomState(6).randn(4100, 256)
    x.flags.writeable = False
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # any warning fails the test
        a, b = engine.upload(x), engine.upload64(x)
    assert torch.equal(b.cpu(), torch.from_numpy(x.copy())) and torch.equal(a, b.float())
    assert not x.flags.writeable
