from .model import LoFTR, PositionEncodingSine  # noqa: F401
from .transformer import LocalFeatureTransformer, LocalFeatureTransformerRegressor  # noqa: F401
from .stages import CoarseMatching, FineMatching, FinePreprocess  # noqa: F401
