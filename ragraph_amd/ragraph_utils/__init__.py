from .Propagation import Propagation  # noqa: F401
from .SimilarityFunctions import SimilarityFunctions  # noqa: F401
from .TaskDecoder import TaskDecoder  # noqa: F401
from .ToyGraphBase import ToyGraphBase  # noqa: F401
from .utility import process_tu_dataset, seed_everything  # noqa: F401
