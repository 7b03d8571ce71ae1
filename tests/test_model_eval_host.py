"""Interface pieces around the scoring path that need no device: the oracle against the reference's
outputs (tests/golden/model_eval.npz), statistics centring, matrix files, command-line helpers."""
import argparse

import numpy as np
import pytest

from conftest import model_from
from oracle import plda_ref as O


def test_oracle_matches_reference_outputs(golden):
    g = golden("model_eval")
    x, y, z = g["x"], g["y"], g["z"]
    sp, pl = model_from(g, "splda_", "splda"), model_from(g, "plda_", "plda")
    assert np.allclose(O.splda_log_probx_g_y(sp, x, y), g["splda_logp_y"], rtol=1e-12)
    assert np.allclose(O.plda_log_probx_g_y(pl, x, y), g["plda_logp_y"], rtol=1e-10)
    assert np.allclose(O.plda_log_probx_g_yz(pl, x, y, z), g["plda_logp_yz"], rtol=1e-12)
    N, F, S = O.center_stats(O.compute_stats_hard(x, g["ids"]), g["cs_mu"])
    assert np.array_equal(N, g["cs_N"]) and np.allclose(F, g["cs_F"]) and np.allclose(S, g["cs_S"])
    for tag, var_r, min_dim, want in g["pca_dim_cases"]:
        xx = g["pca_x_full"] if tag == 0 else g["pca_x_def"]
        assert O.pca_dim_for_var_ratio(xx, var_r if var_r != 1 else 1, int(min_dim)) == int(want)


def test_center_stats_is_host_algebra_and_leaves_inputs_alone(golden):
    from hyperion_amd.pdfs.plda import SPLDA

    g = golden("model_eval")
    D = O.compute_stats_hard(g["x"], g["ids"])
    keep = [a.copy() for a in D]
    N, F, S = SPLDA.center_stats(D, g["cs_mu"])
    assert np.array_equal(N, g["cs_N"])
    assert np.allclose(F, g["cs_F"], rtol=1e-12, atol=1e-12) and np.allclose(S, g["cs_S"], rtol=1e-12, atol=1e-10)
    for a, b in zip(D, keep):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("cls_name", ["CentWhiten", "LDA", "PCA", "LNorm"])
def test_matrix_files_round_trip(tmp_path, cls_name):
    import hyperion_amd.transforms as T

    cls = getattr(T, cls_name)
    rng = np.random.RandomState(3)
    t = cls(mu=rng.randn(7), T=rng.randn(7, 4))
    path = str(tmp_path / "mat.h5")
    t.save_mat(path)
    back = cls.load_mat(path)
    assert type(back) is cls
    # float32 precision on the way in, as the reference reads them
    assert back.mu.dtype == np.float32 and np.array_equal(back.mu, t.mu.astype(np.float32))
    assert np.array_equal(back.T, t.T.astype(np.float32))


def test_abstract_slots_and_json_config():
    from hyperion_amd.hyp_model import HypModel
    from hyperion_amd.pdfs.plda import PLDABase, SPLDA

    m = HypModel(name="m")
    for call in (m.initialize, lambda: m.fit(None), lambda: m.fit_generator(None)):
        with pytest.raises(NotImplementedError):
            call()
    assert HypModel.load_config_from_json(m.to_json()) == {"class_name": "HypModel", "name": "m"}
    with pytest.raises(NotImplementedError):
        PLDABase.sample(SPLDA(), 1, 1)
    with pytest.raises(NotImplementedError):
        SPLDA().weigthed_avg_model(None)


def test_command_line_helpers():
    from hyperion_amd.helpers.plda_factory import PLDAFactory as F
    from hyperion_amd.helpers.trial_data_reader import TrialDataReader
    from hyperion_amd.helpers.vector_class_reader import VectorClassReader
    from hyperion_amd.helpers.vector_reader import VectorReader
    from hyperion_amd.transforms import PCA, CentWhiten

    p = argparse.ArgumentParser()
    F.add_class_args(p)
    a = F.filter_train_args(**vars(p.parse_args(["--plda-type", "plda", "--no-update-U", "--z-dim", "7"])))
    assert a["plda_type"] == "plda" and a["z_dim"] == 7 and a["update_U"] is False and a["update_V"] is True
    assert a["fullcov_W"] is True and "no_update_U" not in a and "diag_W" not in a
    assert a["epochs"] == 40 and a["ml_md"] == "ml+md" and a["md_epochs"] is None and a["y_dim"] == 150
    assert F.filter_train_args(y_dim=3, other=1) == {"y_dim": 3}
    p = argparse.ArgumentParser()
    F.add_eval_args(p, "be")
    ns = vars(p.parse_args(["--be.model-file", "m.h5"]))
    assert ns == {"be.plda_type": "splda", "be.model_file": "m.h5"}
    assert F.filter_eval_args(plda_type="splda", model_file="m.h5", x=1) == {"plda_type": "splda", "model_file": "m.h5"}

    p = argparse.ArgumentParser()
    TrialDataReader.add_class_args(p)
    ns = vars(p.parse_args(["--num-seg-parts", "4", "--seg-part-idx", "2", "--eval-set", "COH-TEST"]))
    assert ns["eval_set"] == "coh-test" and ns["seg_part_idx"] == 2 and ns["model_part_idx"] == 1
    assert TrialDataReader.filter_args(**ns) == {"tlist_sep": " ", "num_model_parts": 1, "num_seg_parts": 4,
                                                 "eval_set": "coh-test"}
    p = argparse.ArgumentParser()
    VectorReader.add_class_args(p)
    assert VectorReader.filter_args(**vars(p.parse_args([]))) == {"vlist_sep": " "}
    p = argparse.ArgumentParser()
    VectorClassReader.add_class_args(p)
    ns = vars(p.parse_args(["--max-spc", "5", "--csplit-mode", "SEQUENTIAL", "--no-csplit-once"]))
    kw = VectorClassReader.filter_args(**ns)
    assert kw["max_spc"] == 5 and kw["csplit_mode"] == "sequential" and kw["vcr_seed"] == 1024 and kw["min_spc"] == 1

    from hyperion_amd.diarization import DiarAHCPLDA

    p = argparse.ArgumentParser()
    DiarAHCPLDA.add_class_args(p)
    assert DiarAHCPLDA.filter_args(**vars(p.parse_args(["--threshold", "-1.5", "--use-bic"]))) == {
        "threshold": -1.5, "pca_var_r": 1, "do_unsup_cal": False, "use_bic": True}

    p = argparse.ArgumentParser()
    PCA.add_class_args(p)
    assert PCA.filter_args(**vars(p.parse_args(["--pca-dim", "3"]))) == {
        "update_mu": True, "update_T": True, "name": "pca", "pca_dim": 3, "pca_var_r": None}
    p = argparse.ArgumentParser()
    CentWhiten.add_argparse_args(p)
    assert CentWhiten.filter_args(**vars(p.parse_args([])), junk=1) == {"update_mu": True, "update_T": True,
                                                                      "name": "lnorm"}


def test_list_and_trial_tables_keep_the_old_comparison_protocol():
    from hyperion_amd.utils.scp_list import SCPList
    from hyperion_amd.utils.trial_ndx import TrialKey, TrialNdx
    from hyperion_amd.utils.utt2info import Utt2Info

    a = SCPList(["u1", "u2"], ["a.ark", "a.ark"], [5, 90])
    b = a.copy()
    assert a.__cmp__(b) == 0
    b.offset = [5, 91]
    b.range_spec = []
    b.validate()
    assert b.offset.dtype == np.int64 and b.range_spec is None and a.__cmp__(b) == 1
    b.offset = [5]
    with pytest.raises(AssertionError):
        b.validate()
    u = Utt2Info.create(["u1", "u2"], ["s1", "s2"])
    u.validate()
    assert u.__cmp__(u.copy()) == 0 and u.__cmp__(Utt2Info.create(["u1", "u2"], ["s1", "s3"])) == 1
    mask = np.array([[True, False], [True, True]])
    n1 = TrialNdx(["m1", "m2"], ["t1", "t2"], mask)
    assert n1.__cmp__(n1.copy()) == 0 and n1.__cmp__(TrialNdx(["m1", "m2"], ["t1", "t2"], ~mask)) == 1
    k = TrialKey(["m1", "m2"], ["t1", "t2"], mask, ~mask)
    assert k.__cmp__(k.copy()) == 0


def test_oracle_roc_calibration_cosine_match_reference(golden):
    g = golden("roc_cal_cos")
    tar, non = g["tar"], g["non"]
    pm, pf = O.compute_roc(tar, non)
    assert np.array_equal(pm, g["roc_pmiss"]) and np.array_equal(pf, g["roc_pfa"])
    for a, k in zip(O.pavx(g["pav_y"]), ("pav_ghat", "pav_width", "pav_height")):
        assert np.array_equal(a, g[k])
    for method in ("laplace", "raw"):
        t, n = O.opt_loglr(tar, non, method)
        assert np.array_equal(t, g["cal_tar_" + method]) and np.array_equal(n, g["cal_non_" + method])
    for d in (512, 37):
        assert np.allclose(O.cosine_scoring(g["cos%d_x1" % d], g["cos%d_x2" % d]), g["cos%d" % d], rtol=0, atol=1e-14)


def test_pavx_is_host_code_and_bit_equal(golden):
    from hyperion_amd.metrics import pavx

    g = golden("roc_cal_cos")
    ghat, width, height = pavx(g["pav_y"])
    assert np.array_equal(ghat, g["pav_ghat"]) and np.array_equal(width, g["pav_width"])
    assert np.array_equal(height, g["pav_height"])
    assert np.all(np.diff(height) > 0) and width.sum() == len(g["pav_y"])
    one = pavx([3.0])
    assert one[0].tolist() == [3.0] and one[1].tolist() == [1]
    with pytest.raises(AssertionError):
        pavx([])


def test_io_factory_options_and_logger_levels():
    import logging

    from hyperion_amd import hyp_defs
    from hyperion_amd.io import DataWriterFactory, RandomAccessDataReaderFactory

    p = argparse.ArgumentParser()
    DataWriterFactory.add_class_args(p)
    RandomAccessDataReaderFactory.add_class_args(p)  # --scp-sep is already there: kept, not an error
    ns = vars(p.parse_args(["--compress", "--compression-method", "2byte-auto", "--path-prefix", "/data"]))
    assert DataWriterFactory.filter_args(**ns) == {"scp_sep": " ", "compress": True, "compression_method": "2byte-auto"}
    assert RandomAccessDataReaderFactory.filter_args(**ns) == {"scp_sep": " ", "path_prefix": "/data"}
    with pytest.raises(SystemExit):
        p.parse_args(["--compression-method", "zip"])
    assert hyp_defs.logging_levels == {0: logging.WARN, 1: logging.INFO, 2: logging.DEBUG, 3: 5}
    assert hyp_defs.float_keras() == "float32"
    hyp_defs.set_float_keras("float16")
    assert hyp_defs.float_keras() == "float16"
    hyp_defs.set_float_keras("float32")
    assert callable(hyp_defs.config_logger)
