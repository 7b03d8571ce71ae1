"""Loader of labelled training vectors for LDA / PLDA training: a vector file, a `key class` list,
optional pruning of the classes by size and splitting of large classes into sub-classes
(hyperion/helpers/vector_class_reader.py:20-236).  The random choices use numpy's RandomState in the
reference's order of calls, so the same seed selects the same vectors."""
import numpy as np

from ..io import RandomAccessDataReaderFactory as DRF
from ..utils.utt2info import Utt2Info


def to3D_by_class(x, class_ids, max_length=0):
    """[num_classes, max_length, dim] array with the vectors of every class in list order, and the
    0/1 weights of the rows that are filled (hyperion/utils/tensors.py:9-25)."""
    class_ids = np.asarray(class_ids)
    num_classes = int(np.max(class_ids)) + 1
    counts = np.bincount(class_ids, minlength=num_classes)
    if max_length == 0:
        max_length = int(counts.max())
    x3d = np.zeros((num_classes, max_length, x.shape[1]), dtype=x.dtype)
    sample_weight = np.zeros((num_classes, max_length), dtype=x.dtype)
    for i in range(num_classes):
        x3d[i, :counts[i], :] = x[class_ids == i, :]
        sample_weight[i, :counts[i]] = 1.0
    return x3d, sample_weight


class VectorClassReader(object):
    def __init__(self, v_file, key_file, preproc=None, vlist_sep=" ", class2int_file=None, min_spc=1,
                 max_spc=None, spc_pruning_mode="random", csplit_min_spc=1, csplit_max_spc=None,
                 csplit_mode="random", csplit_overlap=0, vcr_seed=1024, csplit_once=True):
        self.r = DRF.create(v_file)
        self.u2c = Utt2Info.load(key_file, sep=vlist_sep)
        self.preproc = preproc
        self.map_class2int = None
        if class2int_file is not None:
            with open(class2int_file, "r") as f:
                self.map_class2int = {v[0]: int(v[1]) for v in (line.rstrip().split() for line in f)}
        self.rng = np.random.RandomState(vcr_seed)
        self.csplit_max_spc = csplit_max_spc
        self.csplit_min_spc = csplit_min_spc
        self.csplit_mode = csplit_mode
        self.csplit_overlap = csplit_overlap
        self.csplit_once = csplit_once
        self._samples_per_class = None
        self.u2c = self._filter_by_spc(self.u2c, min_spc, max_spc, spc_pruning_mode, self.rng)
        if csplit_once:
            self.u2c = self._current_split()

    def _current_split(self):
        return self._split_classes(self.u2c, self.csplit_min_spc, self.csplit_max_spc, self.csplit_mode,
                                   self.csplit_overlap, self.rng)

    def _list(self):
        return self.u2c if self.csplit_once else self._current_split()

    def read(self, return_3d=False, max_length=0):
        """-> (x [n, dim], class_ids [n]); with return_3d (x3d, sample_weight) of to3D_by_class.
        The pre-processing chain runs on the device (one folded plan)."""
        u2c = self._list()
        x = self.r.read(u2c.key, squeeze=True)
        if self.preproc is not None:
            x = self.preproc.predict(x)
        if self.map_class2int is None:
            _, class_ids = np.unique(u2c.info, return_inverse=True)
        else:
            class_ids = np.array([self.map_class2int[k] for k in u2c.info], dtype=int)
        if return_3d:
            return to3D_by_class(x, class_ids, max_length)
        return x, class_ids

    @property
    def class_names(self):
        if self.map_class2int is None:
            return np.unique(self.u2c.info)
        by_int = {v: k for k, v in self.map_class2int.items()}
        return np.asarray([by_int[i] for i in range(len(by_int))])

    @property
    def samples_per_class(self):
        if self._samples_per_class is None:
            _, self._samples_per_class = np.unique(self._list().info, return_counts=True)
        return self._samples_per_class

    @property
    def max_samples_per_class(self):
        return np.max(self.samples_per_class)

    @staticmethod
    def _filter_by_spc(u2c, min_spc=1, max_spc=None, spc_pruning_mode="last", rng=None):
        """Drops classes with fewer than min_spc vectors, then cuts every class down to max_spc
        (vector_class_reader.py:130-166): one rng.choice per oversized class, in class order."""
        if min_spc <= 1 and max_spc is None:
            return u2c
        if min_spc > 1:
            classes, num_spc = np.unique(u2c.info, return_counts=True)
            u2c = u2c.filter_info(classes[num_spc >= min_spc])
        if max_spc is not None:
            _, class_ids, num_spc = np.unique(u2c.info, return_inverse=True, return_counts=True)
            if np.all(num_spc <= max_spc):
                return u2c
            keep = np.ones(len(class_ids), dtype=bool)
            for i in np.flatnonzero(num_spc > max_spc):
                indx = np.where(class_ids == i)[0]
                num_reject = len(indx) - max_spc
                if spc_pruning_mode == "random":
                    indx = rng.choice(indx, size=num_reject, replace=False)
                elif spc_pruning_mode == "last":
                    indx = indx[-num_reject:]
                elif spc_pruning_mode == "first":
                    indx = indx[:num_reject]
                keep[indx] = False
            if not np.all(keep):
                u2c = Utt2Info.create(u2c.key[keep], u2c.info[keep])
        return u2c

    @staticmethod
    def _split_classes(u2c, min_spc, max_spc, mode="sequential", overlap=0, rng=None):
        """Classes with more than max_spc vectors become several classes of max_spc vectors: windows
        that advance by max_spc - overlap ("sequential", the last one takes what is left) or
        independent random draws ("random"); the new classes are numbered in order and named by
        their number (vector_class_reader.py:168-228)."""
        if max_spc is None:
            return u2c
        if mode == "random_1part":
            return VectorClassReader._filter_by_spc(u2c, min_spc, max_spc, "random", rng)
        _, class_ids, num_spc = np.unique(u2c.info, return_inverse=True, return_counts=True)
        if np.all(num_spc <= max_spc):
            return VectorClassReader._filter_by_spc(u2c, min_spc)
        shift = max_spc - overlap
        picks, labels = [], []
        for i in range(len(num_spc)):
            indx_i = np.where(class_ids == i)[0]
            parts = [indx_i]
            if num_spc[i] > max_spc:
                num_subclass = int(np.ceil((num_spc[i] - max_spc) / shift + 1))
                if mode == "sequential":
                    starts = [int(k * shift) for k in range(num_subclass - 1)]
                    parts = [indx_i[a:a + max_spc] for a in starts]
                    last = int((num_subclass - 1) * shift)
                    parts.append(indx_i[last:last + int(num_spc[i] - (num_subclass - 1) * shift)])
                elif mode == "random":
                    parts = [rng.choice(indx_i, size=max_spc, replace=False) for _ in range(num_subclass)]
                else:
                    parts = []
            for p in parts:
                picks.append(p)
                labels.append(np.full(len(p), len(labels), dtype=int))
        new_indx = np.concatenate(picks)
        new_class_ids = np.concatenate(labels).astype("U")
        return VectorClassReader._filter_by_spc(Utt2Info.create(u2c.key[new_indx], new_class_ids), min_spc)

    @staticmethod
    def filter_args(**kwargs):
        valid_args = ("vlist_sep", "class2int_file", "min_spc", "max_spc", "spc_pruning_mode", "csplit_min_spc",
                      "csplit_max_spc", "csplit_mode", "csplit_overlap", "csplit_once", "vcr_seed")
        return dict((k, kwargs[k]) for k in valid_args if k in kwargs)
    @staticmethod
    def add_class_args(parser, prefix=None):
        """Options of the training scripts (hyperion/helpers/vector_class_reader.py:248-312)."""
        p1 = "--" if prefix is None else "--" + prefix + "."
        parser.add_argument(p1 + "vlist-sep", default=" ", help="utt2class file field separator")
        parser.add_argument(p1 + "class2int-file", default=None, help="file that maps class string to integer")
        parser.add_argument(p1 + "min-spc", type=int, default=1, help="minimum samples per class")
        parser.add_argument(p1 + "max-spc", type=int, default=None, help="maximum samples per class")
        parser.add_argument(p1 + "spc-pruning-mode", default="random", choices=["random", "first", "last"],
                            help="vector pruning method when spc > max-spc")
        parser.add_argument(p1 + "csplit-min-spc", type=int, default=None,
                            help="minimum samples per class when doing class spliting")
        parser.add_argument(p1 + "csplit-max-spc", type=int, default=None,
                            help="split one class into subclasses with spc <= csplit-max-spc")
        parser.add_argument(p1 + "csplit-mode", default="random", type=str.lower,
                            choices=["sequential", "random", "random_1subclass"], help="class splitting mode")
        parser.add_argument(p1 + "csplit-overlap", type=float, default=0, help="overlap between subclasses")
        parser.add_argument(p1 + "no-csplit-once", default=True, action="store_false",
                            help="class spliting done in each iteration")
        parser.add_argument(p1 + "vcr-seed", type=int, default=1024, help="seed for rng")

    add_argparse_args = add_class_args
