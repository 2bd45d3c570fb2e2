from .loftr import LoFTR  # noqa: F401
from .transformer import LocalFeatureTransformer, LocalFeatureTransformerRegressor  # noqa: F401
from .fine_preprocess import FinePreprocess  # noqa: F401
from .coarse_matching import CoarseMatching  # noqa: F401
from .fine_matching import FineMatching  # noqa: F401
