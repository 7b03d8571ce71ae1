from .trial_data_reader import TrialDataReader  # noqa: F401
from .vector_reader import VectorReader  # noqa: F401
from .vector_class_reader import VectorClassReader  # noqa: F401
from .plda_factory import PLDAFactory  # noqa: F401
