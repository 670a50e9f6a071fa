"""Loss factory — same signature as ``cellulus/criterions/__init__.py:4-17``."""

from .oce_loss import OCELoss


def get_loss(
    temperature,
    regularizer_weight,
    density,
    num_spatial_dims,
    device,
):
    return OCELoss(
        temperature,
        regularizer_weight,
        density,
        num_spatial_dims,
        device,
    )
