from .trial_data_reader import TrialDataReader  # noqa: F401
from .vector_reader import VectorReader  # noqa: F401
