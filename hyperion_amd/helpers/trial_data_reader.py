"""Loads the trial index, the enrolment list and the x-vectors of one evaluation job
(hyperion/helpers/trial_data_reader.py)."""
from ..io import RandomAccessDataReaderFactory as DRF
from ..utils import TrialKey, TrialNdx, Utt2Info


class TrialDataReader(object):
    """Same arguments as the reference (trial_data_reader.py:24-61).  The job grid
    (model_part_idx of num_model_parts, seg_part_idx of num_seg_parts) selects one block of the
    trial matrix; on a multi-GPU node a rank passes its own seg_part_idx to own a shard of tests."""

    def __init__(self, v_file, ndx_file, enroll_file, test_file=None, preproc=None, model_part_idx=1,
                 num_model_parts=1, seg_part_idx=1, num_seg_parts=1, eval_set="enroll-test", tlist_sep=" "):
        self.r = DRF.create(v_file)
        self.preproc = preproc
        enroll = Utt2Info.load(enroll_file, sep=tlist_sep)
        test = Utt2Info.load(test_file, sep=tlist_sep) if test_file is not None else None
        ndx = None
        if ndx_file is not None:
            ndx = self._load_ndx(ndx_file)
        ndx, enroll = TrialNdx.parse_eval_set(ndx, enroll, test, eval_set)
        if num_model_parts > 1 or num_seg_parts > 1:
            ndx = ndx.split(model_part_idx, num_model_parts, seg_part_idx, num_seg_parts)
            enroll = enroll.filter_info(ndx.model_set)
        self.enroll = enroll
        self.ndx = ndx

    @staticmethod
    def _load_ndx(ndx_file):
        """A two-column trial list, or a key file (three columns) reduced to its index
        (trial_data_reader.py:46-51)."""
        if ndx_file.endswith((".h5", ".hdf5")):
            try:
                return TrialNdx.load(ndx_file)
            except KeyError:  # a key file stores +1 / -1 labels under the same dataset names
                return TrialKey.load(ndx_file).to_ndx()
        with open(ndx_file, "r") as f:
            fields = len(f.readline().split())
        return TrialNdx.load(ndx_file) if fields == 2 else TrialKey.load(ndx_file).to_ndx()

    def read(self):
        """x_e [enrol segments, d], x_t [test segments, d] (after preproc.predict when a chain was
        given), the model name of every enrolment segment, and the trial index
        (trial_data_reader.py:63-72)."""
        x_e = self.r.read(self.enroll.key, squeeze=True)
        x_t = self.r.read(self.ndx.seg_set, squeeze=True)
        if self.preproc is not None:
            x_e = self.preproc.predict(x_e)
            x_t = self.preproc.predict(x_t)
        return x_e, x_t, self.enroll.info, self.ndx

    def read_device(self):
        """As read(), but the raw vectors stay on the device (page-locked staging, one transfer per
        side) and the chain is NOT applied: hand `preproc` to llr_1vs1(..., preproc=) or to
        PLDABackend, which fold it into the embedding prologue."""
        return (self.r.read_device(self.enroll.key), self.r.read_device(self.ndx.seg_set), self.enroll.info,
                self.ndx)

    @staticmethod
    def filter_args(**kwargs):
        valid = ("tlist_sep", "model_idx", "num_model_parts", "seg_idx", "num_seg_parts", "eval_set")
        return dict((k, kwargs[k]) for k in valid if k in kwargs)
    @staticmethod
    def add_class_args(parser, prefix=None):
        """Options of the evaluation scripts (hyperion/helpers/trial_data_reader.py:86-130)."""
        p1 = "--" if prefix is None else "--" + prefix + "."
        parser.add_argument(p1 + "tlist-sep", default=" ", help="trial lists field separator")
        parser.add_argument(p1 + "model-part-idx", default=1, type=int, help="model part index")
        parser.add_argument(p1 + "num-model-parts", default=1, type=int,
                            help="number of parts in which we divide the model list to run evaluation in parallel")
        parser.add_argument(p1 + "seg-part-idx", default=1, type=int, help="test part index")
        parser.add_argument(p1 + "num-seg-parts", default=1, type=int,
                            help="number of parts in which we divide the test list to run evaluation in parallel")
        parser.add_argument(p1 + "eval-set", type=str.lower, default="enroll-test",
                            choices=["enroll-test", "enroll-coh", "coh-test", "coh-coh"], help="evaluation subset")

    add_argparse_args = add_class_args
