"""Score table of a trial set (hyperion/utils/trial_scores.py): scores and the mask of trials that
have one, over models x test segments."""
import numpy as np

from ..hyp_defs import float_cpu
from .list_utils import split_list
from .trial_ndx import TrialNdx, TrialTable, _h5_writer, _ids_and_positions, _is_h5, _open_h5, _read_columns


class TrialScores(TrialTable):
    PLANES = ("scores", "score_mask")

    def __init__(self, model_set=None, seg_set=None, scores=None, score_mask=None):
        self._init_sets(model_set, seg_set)
        self.scores = scores
        self.score_mask = score_mask
        if model_set is not None and seg_set is not None:
            self.validate()

    def validate(self):
        """Unique names, finite scores; missing scores are zeros, a missing mask is all-true
        (trial_scores.py:289-307)."""
        self._validate_sets()
        if self.scores is None:
            self.scores = np.zeros(self.shape)
        else:
            assert self.scores.shape == self.shape
            assert np.all(np.isfinite(self.scores))
        if self.score_mask is None:
            self.score_mask = np.ones(self.shape, dtype=bool)
        else:
            assert self.score_mask.shape == self.shape

    def _rebuild(self, model_set, seg_set, planes):
        return TrialScores(model_set, seg_set, planes["scores"], planes["score_mask"])

    def save(self, file_path):
        (self.save_h5 if _is_h5(file_path) else self.save_txt)(file_path)

    def save_h5(self, file_path):
        with _h5_writer(file_path) as f:
            f.create_dataset("ID/row_ids", data=self.model_set.astype("S"))
            f.create_dataset("ID/column_ids", data=self.seg_set.astype("S"))
            f.create_dataset("scores", data=self.scores)
            f.create_dataset("score_mask", data=self.score_mask.astype("uint8"))

    def save_txt(self, file_path, block=1 << 16):
        """`model segment score` with %f, segment-major over the trials that have a score
        (trial_scores.py:85-101); written in blocks of joined lines."""
        si, mi = self.score_mask.T.nonzero()
        with open(file_path, "w") as f:
            for b in range(0, len(si), block):
                m, s = mi[b:b + block], si[b:b + block]
                f.write("".join("%s %s %f\n" % t for t in zip(self.model_set[m], self.seg_set[s],
                                                             self.scores[m, s])))

    @classmethod
    def load(cls, file_path):
        return cls.load_h5(file_path) if _is_h5(file_path) else cls.load_txt(file_path)

    @classmethod
    def load_h5(cls, file_path):
        f = _open_h5(file_path)
        model_set = [t.decode("utf-8") for t in f["ID/row_ids"]]
        seg_set = [t.decode("utf-8") for t in f["ID/column_ids"]]
        scores = np.asarray(f["scores"], dtype=float_cpu())
        score_mask = np.asarray(f["score_mask"], dtype=bool)
        return cls(model_set, seg_set, scores, score_mask)

    @classmethod
    def load_txt(cls, file_path):
        """Lines `model segment score` (trial_scores.py:136-164)."""
        models, segments, values = _read_columns(file_path, 3)
        model_set, mi = _ids_and_positions(models)
        seg_set, si = _ids_and_positions(segments)
        scores = np.zeros((len(model_set), len(seg_set)))
        mask = np.zeros(scores.shape, dtype=bool)
        scores[mi, si] = np.asarray(values, dtype=np.float64)
        mask[mi, si] = True
        return cls(model_set, seg_set, scores, mask)

    @classmethod
    def merge(cls, scr_list):
        """Union of score tables that do not overlap (trial_scores.py:166-222)."""
        counts = None
        for t in scr_list:
            one = cls(t.model_set, t.seg_set, t.score_mask.astype(np.float64), t.score_mask)
            counts = [one] if counts is None else counts + [one]
        model_set, seg_set, planes = cls._merge_planes(scr_list)
        _, _, times = cls._merge_planes(counts)
        assert not np.any(times["scores"] > 1), "two tables score the same trial"
        return cls(model_set, seg_set, planes["scores"], planes["score_mask"])

    def filter(self, model_set, seg_set, keep=True, raise_missing=True):
        """Sub-table in the order given; names that are not in the table give unscored rows /
        columns, or an exception when raise_missing (trial_scores.py:224-266)."""
        model_set, seg_set, fm, mi, fs, si = self._locate(model_set, seg_set, keep)
        if np.all(fm) and np.all(fs):
            return self._take(model_set, seg_set, mi, si)
        if raise_missing:
            raise Exception("some scores were not computed")
        scores = np.zeros((len(model_set), len(seg_set)), dtype=float_cpu())
        mask = np.zeros(scores.shape, dtype=bool)
        dst, src = np.ix_(fm, fs), np.ix_(mi[fm], si[fs])
        scores[dst] = self.scores[src]
        mask[dst] = self.score_mask[src]
        return TrialScores(model_set, seg_set, scores, mask)

    def align_with_ndx(self, ndx, raise_missing=True):
        """Rows / columns in the order of `ndx` (a TrialNdx or TrialKey), scored trials limited to
        those the index asks for (trial_scores.py:309-340)."""
        scr = self.filter(ndx.model_set, ndx.seg_set, keep=True, raise_missing=raise_missing)
        wanted = ndx.trial_mask if isinstance(ndx, TrialNdx) else np.logical_or(ndx.tar, ndx.non)
        missing = np.logical_and(wanted, np.logical_not(scr.score_mask))
        scr.score_mask = np.logical_and(wanted, scr.score_mask)
        if raise_missing and np.any(missing):
            raise Exception("some scores were not computed")
        return scr

    def get_tar_non(self, key):
        """Target and non-target score vectors, row-major over the key (trial_scores.py:342-357)."""
        scr = self.align_with_ndx(key)
        return (scr.scores[np.logical_and(scr.score_mask, key.tar)],
                scr.scores[np.logical_and(scr.score_mask, key.non)])

    def set_missing_to_value(self, ndx, val):
        """Trials the index asks for but that have no score get `val` (trial_scores.py:359-378)."""
        scr = self.align_with_ndx(ndx, raise_missing=False)
        wanted = ndx.trial_mask if isinstance(ndx, TrialNdx) else np.logical_or(ndx.tar, ndx.non)
        fill = np.logical_and(np.logical_not(scr.score_mask), wanted)
        scr.scores[fill] = val
        scr.score_mask[fill] = True
        return scr

    def transform(self, f):
        """Applies f to the scored entries in place (trial_scores.py:380-387)."""
        self.scores[self.score_mask] = f(self.scores[self.score_mask])

    def __eq__(self, other):
        return bool(self._same_sets(other) and np.all(np.isclose(self.scores, other.scores, atol=1e-5)) and
                    np.all(self.score_mask == other.score_mask))
