"""The diarization consumer on the device (pre-processing, recording-level PCA, projected SPLDA,
self-scoring) against the reference's cluster ids and linkage (tests/golden/diarization.npz)."""
import numpy as np
import pytest

from conftest import model_from
from oracle import plda_ref as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,var_r,thr", (("full", 1, 4.0), ("pca", 0.8, 2.0)))
def test_cluster_matches_reference(golden, tag, var_r, thr):
    from hyperion_amd.diarization import DiarAHCPLDA
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LNorm

    g = golden("diarization")
    m = model_from(g, "m_", "splda")
    diar = DiarAHCPLDA(SPLDA(mu=m["mu"], V=m["V"], W=m["W"]), preproc=LNorm(), threshold=thr, pca_var_r=var_r)
    ids = diar.cluster(g["x"])
    assert np.array_equal(ids, g["ids_" + tag])
    ref = g["Z_" + tag]
    assert np.array_equal(diar._ahc.Z[:, :2], ref[:, :2]) and np.array_equal(diar._ahc.Z[:, 3], ref[:, 3])
    assert np.allclose(diar._ahc.Z[:, 2], ref[:, 2], rtol=1e-4, atol=1e-3)


def test_recording_with_many_segments():
    """2000 segments of 6 speakers in 128 dimensions: same partition as the float64 oracle."""
    from hyperion_amd.diarization import DiarAHCPLDA
    from hyperion_amd.pdfs.plda import SPLDA
    from hyperion_amd.transforms import LNorm

    import synth

    model = synth.make_splda(128, 40)
    rng = np.random.RandomState(4)
    spk = rng.randint(0, 6, 2000)
    x = model["mu"] + (1.5 * rng.randn(6, 40))[spk] @ model["V"] + rng.randn(2000, 128) @ np.linalg.cholesky(
        np.linalg.inv(model["W"])).T
    diar = DiarAHCPLDA(SPLDA(mu=model["mu"], V=model["V"], W=model["W"]), preproc=LNorm(), threshold=0.0,
                       pca_var_r=0.9)
    ids = diar.cluster(x)
    want, _ = O.diar_ahc_plda_cluster(model, x, 0.0, 0.9)
    assert np.array_equal(ids, want)
    with pytest.raises(NotImplementedError):
        DiarAHCPLDA(None, do_unsup_cal=True)
