"""Trial index, key and score tables: models x test segments (Bosaris-style Ndx / Key / Scores,
hyperion/utils/trial_ndx.py, trial_key.py, trial_scores.py).

The three classes share one table: two id lists plus named [num_models, num_tests] planes.  Text
files are parsed with array operations (np.unique + one fancy assignment) instead of a Python loop
per trial."""
import copy
import os.path as path

import numpy as np

from .list_utils import intersect, ismember, list2ndarray, sort, split_list


def _is_h5(file_path):
    return path.splitext(file_path)[1] in (".h5", ".hdf5")


def _open_h5(file_path):
    from ..io.hdf5_lite import Hdf5File

    return Hdf5File(file_path)


def _h5_writer(file_path):
    """An object with h5py's `create_dataset(name, data=...)`, from the package's own HDF5 writer."""
    from ..io.hdf5_write import File

    return File(file_path, "w")


def _read_columns(file_path, num_fields):
    cols = [[] for _ in range(num_fields)]
    with open(file_path, "r") as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            for c in range(num_fields):
                cols[c].append(parts[c])
    return cols


def _ids_and_positions(names):
    ids, pos = np.unique(np.asarray(names, dtype=str), return_inverse=True)
    return ids, pos


class TrialTable(object):
    """model_set / seg_set and the planes named in PLANES (attribute name -> dtype).  Sub-classes
    say which planes they carry and what a missing plane defaults to."""

    PLANES = ()

    def _init_sets(self, model_set, seg_set):
        self.model_set = model_set
        self.seg_set = seg_set

    @property
    def num_models(self):
        return len(self.model_set)

    @property
    def num_tests(self):
        return len(self.seg_set)

    @property
    def shape(self):
        return (len(self.model_set), len(self.seg_set))

    def copy(self):
        return copy.deepcopy(self)

    def __cmp__(self, other):
        """0 when equal, 1 otherwise (trial_ndx.py:321-325, trial_key.py:514-518,
        trial_scores.py:403-407; two of the three name an undefined variable there)."""
        return 0 if self.__eq__(other) else 1

    def _validate_sets(self):
        self.model_set = list2ndarray(self.model_set)
        self.seg_set = list2ndarray(self.seg_set)
        assert len(np.unique(self.model_set)) == len(self.model_set), "model names repeat"
        assert len(np.unique(self.seg_set)) == len(self.seg_set), "segment names repeat"

    def _planes(self):
        return {n: getattr(self, n) for n in self.PLANES}

    def _rebuild(self, model_set, seg_set, planes):
        raise NotImplementedError

    def _take(self, model_set, seg_set, mi, si):
        ix = np.ix_(mi, si)
        return self._rebuild(model_set, seg_set, {n: p[ix] for n, p in self._planes().items()})

    def sort(self):
        """Orders models and segments by name (trial_ndx.py:43-47)."""
        self.model_set, mi = sort(self.model_set, return_index=True)
        self.seg_set, si = sort(self.seg_set, return_index=True)
        ix = np.ix_(mi, si)
        for n, p in self._planes().items():
            setattr(self, n, p[ix])
        return mi, si

    def split(self, model_idx, num_model_parts, seg_idx, num_seg_parts):
        """Block (model_idx, seg_idx), both 1-based, of the num_model_parts x num_seg_parts grid of
        contiguous parts (trial_ndx.py:252-268)."""
        model_set, mi = split_list(self.model_set, model_idx, num_model_parts)
        seg_set, si = split_list(self.seg_set, seg_idx, num_seg_parts)
        return self._take(model_set, seg_set, mi, si)

    def _locate(self, model_set, seg_set, keep):
        if not keep:
            model_set = np.setdiff1d(self.model_set, model_set)
            seg_set = np.setdiff1d(self.seg_set, seg_set)
        fm, mi = ismember(model_set, self.model_set)
        fs, si = ismember(seg_set, self.seg_set)
        return np.asarray(model_set), np.asarray(seg_set), fm, mi, fs, si

    def filter(self, model_set, seg_set, keep=True):
        """The sub-table of the given models and segments, in the order given (keep), or of all
        the others (trial_ndx.py:227-250); every name must exist."""
        model_set, seg_set, fm, mi, fs, si = self._locate(model_set, seg_set, keep)
        assert np.all(fm) and np.all(fs), "model or segment not in the table"
        return self._take(model_set, seg_set, mi, si)

    @classmethod
    def _merge_planes(cls, tables):
        """Union of the id sets (sorted) with every plane OR-ed / summed into place
        (trial_ndx.py:151-195)."""
        model_set = tables[0].model_set
        seg_set = tables[0].seg_set
        for t in tables[1:]:
            model_set = np.union1d(model_set, t.model_set)
            seg_set = np.union1d(seg_set, t.seg_set)
        shape = (len(model_set), len(seg_set))
        merged = {n: np.zeros(shape, dtype=tables[0]._planes()[n].dtype) for n in cls.PLANES}
        for t in tables:
            _, mi_new, mi_old = intersect(model_set, t.model_set, assume_unique=True, return_index=True)
            _, si_new, si_old = intersect(seg_set, t.seg_set, assume_unique=True, return_index=True)
            new, old = np.ix_(mi_new, si_new), np.ix_(mi_old, si_old)
            for n, plane in t._planes().items():
                if merged[n].dtype == bool:
                    merged[n][new] |= plane[old]
                else:
                    merged[n][new] += plane[old]
        return model_set, seg_set, merged

    def _same_sets(self, other):
        return (self.model_set.shape == other.model_set.shape and np.all(self.model_set == other.model_set) and
                self.seg_set.shape == other.seg_set.shape and np.all(self.seg_set == other.seg_set))

    def __ne__(self, other):
        return not self.__eq__(other)


class TrialNdx(TrialTable):
    """trial_mask[m, s]: the trial (model m, segment s) is to be scored (trial_ndx.py:15-29)."""

    PLANES = ("trial_mask",)

    def __init__(self, model_set=None, seg_set=None, trial_mask=None):
        self._init_sets(model_set, seg_set)
        self.trial_mask = trial_mask
        if model_set is not None and seg_set is not None:
            self.validate()

    def validate(self):
        """Unique names; a missing mask means all trials (trial_ndx.py:270-282)."""
        self._validate_sets()
        if self.trial_mask is None:
            self.trial_mask = np.ones(self.shape, dtype=bool)
        else:
            assert self.trial_mask.shape == self.shape

    def _rebuild(self, model_set, seg_set, planes):
        return TrialNdx(model_set, seg_set, planes["trial_mask"])

    def save(self, file_path):
        (self.save_h5 if _is_h5(file_path) else self.save_txt)(file_path)

    def save_h5(self, file_path):
        """Datasets ID/row_ids, ID/column_ids, trial_mask (trial_ndx.py:61-72)."""
        with _h5_writer(file_path) as f:
            f.create_dataset("ID/row_ids", data=self.model_set.astype("S"))
            f.create_dataset("ID/column_ids", data=self.seg_set.astype("S"))
            f.create_dataset("trial_mask", data=self.trial_mask.astype("uint8"))

    def save_txt(self, file_path):
        """`model segment` per trial, segment-major (trial_ndx.py:83-92)."""
        si, mi = self.trial_mask.T.nonzero()
        with open(file_path, "w") as f:
            f.writelines("%s %s\n" % (m, s) for m, s in zip(self.model_set[mi], self.seg_set[si]))

    @classmethod
    def load(cls, file_path):
        return cls.load_h5(file_path) if _is_h5(file_path) else cls.load_txt(file_path)

    @classmethod
    def load_h5(cls, file_path):
        """trial_ndx.py:110-124, through the package's HDF5 reader."""
        f = _open_h5(file_path)
        model_set = [t.decode("utf-8") for t in f["ID/row_ids"]]
        seg_set = [t.decode("utf-8") for t in f["ID/column_ids"]]
        trial_mask = np.asarray(f["trial_mask"], dtype=bool)
        return cls(model_set, seg_set, trial_mask)

    @classmethod
    def load_txt(cls, file_path):
        """Lines `model segment`; the id sets come out sorted (trial_ndx.py:126-149)."""
        models, segments = _read_columns(file_path, 2)
        model_set, mi = _ids_and_positions(models)
        seg_set, si = _ids_and_positions(segments)
        mask = np.zeros((len(model_set), len(seg_set)), dtype=bool)
        mask[mi, si] = True
        return cls(model_set, seg_set, mask)

    @classmethod
    def merge(cls, ndx_list):
        model_set, seg_set, planes = cls._merge_planes(ndx_list)
        return cls(model_set, seg_set, planes["trial_mask"])

    @staticmethod
    def parse_eval_set(ndx, enroll, test=None, eval_set="enroll-test"):
        """Which models meet which segments for the four evaluation subsets
        (trial_ndx.py:197-225).  For the cohort subsets the segment names are `test.file_path` when
        the list has one (an SCPList) and its keys otherwise (the reference reads `.file_path` of a
        Utt2Info there, which does not exist)."""
        coh_segments = None if test is None else getattr(test, "file_path", test.key)
        if eval_set == "enroll-test":
            enroll = enroll.filter_info(ndx.model_set)
        elif eval_set == "enroll-coh":
            ndx = TrialNdx(ndx.model_set, coh_segments)
            enroll = enroll.filter_info(ndx.model_set)
        elif eval_set == "coh-test":
            ndx = TrialNdx(enroll.key, ndx.seg_set)
        elif eval_set == "coh-coh":
            ndx = TrialNdx(enroll.key, coh_segments)
        return ndx, enroll

    def apply_segmentation_to_test(self, segment_list):
        """Every test file replaced by its sub-segments, trial columns repeated
        (trial_ndx.py:284-306)."""
        per_file = [segment_list.ext_segment_ids_from_file(f) for f in self.seg_set]
        counts = np.array([len(s) for s in per_file])
        return TrialNdx(self.model_set, np.concatenate(per_file), np.repeat(self.trial_mask, counts, axis=1))

    def __eq__(self, other):
        return bool(self._same_sets(other) and np.all(self.trial_mask == other.trial_mask))


class TrialKey(TrialTable):
    """tar / non [num_models, num_tests]: target and non-target trials; optional condition arrays
    (trial_key.py:16-57)."""

    PLANES = ("tar", "non")

    def __init__(self, model_set=None, seg_set=None, tar=None, non=None, model_cond=None, seg_cond=None,
                 trial_cond=None, model_cond_name=None, seg_cond_name=None, trial_cond_name=None):
        self._init_sets(model_set, seg_set)
        self.tar, self.non = tar, non
        self.model_cond, self.seg_cond, self.trial_cond = model_cond, seg_cond, trial_cond
        self.model_cond_name, self.seg_cond_name, self.trial_cond_name = \
            model_cond_name, seg_cond_name, trial_cond_name
        if model_set is not None and seg_set is not None:
            self.validate()

    def validate(self):
        """trial_key.py:446-474."""
        self._validate_sets()
        if self.tar is None or self.non is None:
            self.tar = np.zeros(self.shape, dtype=bool)
            self.non = np.zeros(self.shape, dtype=bool)
        else:
            assert self.tar.shape == self.shape and self.non.shape == self.shape
        if self.model_cond is not None:
            assert self.model_cond.shape[1] == self.shape[0]
        if self.seg_cond is not None:
            assert self.seg_cond.shape[1] == self.shape[1]
        if self.trial_cond is not None:
            assert self.trial_cond.shape[1:] == self.shape
        for n in ("model_cond_name", "seg_cond_name", "trial_cond_name"):
            if getattr(self, n) is not None:
                setattr(self, n, list2ndarray(getattr(self, n)))

    def _rebuild(self, model_set, seg_set, planes):
        return TrialKey(model_set, seg_set, planes["tar"], planes["non"])

    def _take(self, model_set, seg_set, mi, si):
        key = super()._take(model_set, seg_set, mi, si)
        if self.model_cond is not None:
            key.model_cond = self.model_cond[:, mi]
        if self.seg_cond is not None:
            key.seg_cond = self.seg_cond[:, si]
        if self.trial_cond is not None:
            key.trial_cond = self.trial_cond[:, mi][:, :, si]
        key.model_cond_name, key.seg_cond_name, key.trial_cond_name = \
            self.model_cond_name, self.seg_cond_name, self.trial_cond_name
        return key

    def sort(self):
        mi, si = super().sort()
        if self.model_cond is not None:
            self.model_cond = self.model_cond[:, mi]
        if self.seg_cond is not None:
            self.seg_cond = self.seg_cond[:, si]
        if self.trial_cond is not None:
            self.trial_cond = self.trial_cond[:, mi][:, :, si]

    def save(self, file_path):
        (self.save_h5 if _is_h5(file_path) else self.save_txt)(file_path)

    def save_h5(self, file_path):
        """trial_mask = +1 target, -1 non-target (trial_key.py:98-126)."""
        with _h5_writer(file_path) as f:
            f.create_dataset("ID/row_ids", data=self.model_set.astype("S"))
            f.create_dataset("ID/column_ids", data=self.seg_set.astype("S"))
            f.create_dataset("trial_mask", data=self.tar.astype("int8") - self.non.astype("int8"))
            for n in ("model_cond", "seg_cond", "trial_cond"):
                if getattr(self, n) is not None:
                    f.create_dataset(n, data=getattr(self, n).astype("uint8"))
            for n in ("model_cond_name", "seg_cond_name", "trial_cond_name"):
                if getattr(self, n) is not None:
                    f.create_dataset(n, data=getattr(self, n).astype("S"))

    def save_txt(self, file_path):
        """All targets, then all non-targets, `model segment target|nontarget`, segment-major
        (trial_key.py:128-145)."""
        with open(file_path, "w") as f:
            for plane, label in ((self.tar, "target"), (self.non, "nontarget")):
                si, mi = plane.T.nonzero()
                f.writelines("%s %s %s\n" % (m, s, label) for m, s in zip(self.model_set[mi], self.seg_set[si]))

    @classmethod
    def load(cls, file_path):
        return cls.load_h5(file_path) if _is_h5(file_path) else cls.load_txt(file_path)

    @classmethod
    def load_h5(cls, file_path):
        """trial_key.py:163-211, through the package's HDF5 reader."""
        f = _open_h5(file_path)
        model_set = [t.decode("utf-8") for t in f["ID/row_ids"]]
        seg_set = [t.decode("utf-8") for t in f["ID/column_ids"]]
        mask = np.asarray(f["trial_mask"], dtype="int8")
        opt = {n: (np.asarray(f[n], dtype=bool) if n in f else None)
               for n in ("model_cond", "seg_cond", "trial_cond")}
        names = {n: (np.asarray(f[n], dtype="U") if n in f else None)
                 for n in ("model_cond_name", "seg_cond_name", "trial_cond_name")}
        return cls(model_set, seg_set, mask > 0, mask < 0, opt["model_cond"], opt["seg_cond"],
                   opt["trial_cond"], names["model_cond_name"], names["seg_cond_name"], names["trial_cond_name"])

    @classmethod
    def load_txt(cls, file_path):
        """Lines `model segment target|nontarget` (anything but `target` is a non-target,
        trial_key.py:213-241)."""
        models, segments, labels = _read_columns(file_path, 3)
        model_set, mi = _ids_and_positions(models)
        seg_set, si = _ids_and_positions(segments)
        is_tar = np.asarray(labels, dtype=str) == "target"
        tar = np.zeros((len(model_set), len(seg_set)), dtype=bool)
        non = np.zeros_like(tar)
        tar[mi[is_tar], si[is_tar]] = True
        non[mi[~is_tar], si[~is_tar]] = True
        return cls(model_set, seg_set, tar, non)

    @classmethod
    def merge(cls, key_list):
        """Union of several keys (trial_key.py:243-341); condition arrays are not merged."""
        model_set, seg_set, planes = cls._merge_planes(key_list)
        return cls(model_set, seg_set, planes["tar"], planes["non"])

    def to_ndx(self):
        """tar | non as a TrialNdx (trial_key.py:437-444)."""
        return TrialNdx(self.model_set, self.seg_set, np.logical_or(self.tar, self.non))

    def class_keys(self):
        """When every model meets every segment and targets form blocks (each model and segment
        belongs to one class): integer class ids (key_model, key_seg) with tar[m, s] ==
        (key_model[m] == key_seg[s]), else None.  Lets a score sink label trials without the two
        [models x segments] byte matrices."""
        if not np.all(self.tar | self.non):
            return None
        rows, row_class = np.unique(self.tar, axis=0, return_inverse=True)
        row_class = np.asarray(row_class).reshape(-1)
        if np.any(rows.sum(axis=0) > 1):
            return None  # a segment is a target of two different row patterns
        # a segment that is nobody's target gets a class of its own (ids stay non-negative: a negative
        # id would mean "not a trial" to the sink)
        seg_class = np.where(rows.any(axis=0), rows.argmax(axis=0), len(rows) + np.arange(rows.shape[1]))
        ok = np.array_equal(row_class[:, None] == seg_class[None, :], self.tar)
        return (row_class.astype(np.int32), seg_class.astype(np.int32)) if ok else None

    def __eq__(self, other):
        eq = bool(self._same_sets(other) and np.all(self.tar == other.tar) and np.all(self.non == other.non))
        for n in ("model_cond", "seg_cond", "trial_cond", "model_cond_name", "seg_cond_name", "trial_cond_name"):
            a, b = getattr(self, n), getattr(other, n)
            eq = eq and (a is None) == (b is None) and (a is None or bool(np.all(a == b)))
        return eq
