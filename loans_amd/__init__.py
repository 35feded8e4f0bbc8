"""loans_amd -- MI355X-native LoANs localizer + assessor training hot path.

Public surface mirrors the reference's (Bartzi/loans) for this path:
``SheepLocalizer``, ``ResnetAssessor``, ``SheepAssessor`` (alias ``SheepUpdater``),
``rotation_dropout``, ``DirectionLossCalculator``, ``OutOfImageLossCalculator``, ``Adam``.
"""
from .runtime.core import (Variable, Function, Link, Chain, ChainList, Parameter, config, using_config,  # noqa: F401
                           report, reporter, save_npz, load_npz)
from .runtime.optimizers import Adam  # noqa: F401
from .sheep.sheep_localizer import SheepLocalizer, Resnet50SheepLocalizer  # noqa: F401
from .iou.iou_regressor import MyResNet50Layers  # noqa: F401
from .sheep.sheep_updater import SheepAssessor, SheepUpdater  # noqa: F401
from .sheep.sheep_evaluator import SheepMAPEvaluator  # noqa: F401
from .common.net import ResnetAssessor  # noqa: F401
from .common.utils import Size, DirectionLossCalculator, OutOfImageLossCalculator  # noqa: F401
from .functions.rotation_dropout import rotation_dropout, RotationDropout  # noqa: F401

__version__ = '0.1.0'
from .ops import set_compute_dtype, set_storage_dtype  # noqa: F401,E402
