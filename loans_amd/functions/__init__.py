from .basic import add, reshape  # noqa: F401
from .rotation_dropout import rotation_dropout, RotationDropout  # noqa: F401
from .ops_small import (global_average_pooling_2d, linear, spatial_transformer_grid,  # noqa: F401
                        spatial_transformer_sampler, mean_squared_error, sigmoid_linear_head)
