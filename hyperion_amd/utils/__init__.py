from .scp_list import SCPList  # noqa: F401
from .trial_ndx import TrialKey, TrialNdx  # noqa: F401
from .trial_scores import TrialScores  # noqa: F401
from .utt2info import Utt2Info  # noqa: F401
