"""sibrar---single-branch-recommender_amd — MI355X (gfx950) engine for the SiBraR SingleBranchNet hot path.

Hand-written HIP kernels behind a C ABI (include/sibrar_hip.h, csrc/), driven through the reference's own plugin surface:
``SingleBranchNet`` / ``SGDBaseline`` (SGDBasedRecommenderAlgorithm), the ``RecommenderSystemLoss`` classes, ``Trainer`` and
``evaluate_recommender_algorithm`` / ``FullEvaluator``. Import as ``import sibrar_amd`` (root-level shim) or through
``importlib.import_module('sibrar---single-branch-recommender_amd')``.
"""
from ._lib import SibrarHipError, lib, LIB_PATH                                            # noqa: F401
from .config import (DropoutNetConfig, DropoutNetEntityConfig, DropoutNetSamplingStrategy,   # noqa: F401
                     EmbeddingRegularizationType, FeatureModuleConfig, SingleBranchFeatureConfig,
                     SingleBranchNetConfig, SingleBranchNetEntityConfig)
from .features import DeviceTable, HostFeature                                              # noqa: F401
from .polylinear import PolyLinear                                                          # noqa: F401
from .sbnet import (FeatureEmbedding, ItemFeatureMatrixFactorization, SGDBasedRecommenderAlgorithm, SGDBaseline,   # noqa: F401
                    SGDMatrixFactorization, SingleBranchNet, SingleBranchNetEntity, UserFeatureMatrixFactorization,
                    general_weight_init)
from .dropoutnet import DropoutNet, DropoutNetEntity                                      # noqa: F401
from .losses import (InfoNCE, RecBayesianPersonalizedRankingLoss, RecBinaryCrossEntropy,   # noqa: F401
                     RecSampledSoftmaxLoss, RecommenderSystemLoss, RecommenderSystemLossesEnum)
from .optim import FlatParameters, FusedOptimizer                                           # noqa: F401
from .trainer import Trainer                                                                # noqa: F401
from .engine import FusedTrainStep                                                          # noqa: F401
from .evaluation import FullEvaluator, evaluate_recommender_algorithm                       # noqa: F401
from .datasets import NegativeSamplingDataLoader, SyntheticDataset                          # noqa: F401
from .splitdata import SplitDataset, load_split_dataset                                     # noqa: F401
from . import ops, parallel, sampling                                                       # noqa: F401

# the reference's registry: AlgorithmsEnum.sbnet / .sgdbias / .mf -> class (algorithms/algorithms_utils.py:14,17,36)
ALGORITHMS = {'sbnet': SingleBranchNet, 'sgdbias': SGDBaseline, 'mf': SGDMatrixFactorization,
              'ifeatmf': ItemFeatureMatrixFactorization, 'ufeatmf': UserFeatureMatrixFactorization, 'dropoutnet': DropoutNet}
