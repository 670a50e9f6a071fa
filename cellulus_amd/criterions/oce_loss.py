"""``OCELoss`` and the embedding gather on libclx.

Drop-in for ``cellulus/criterions/oce_loss.py`` (forward returns
``(loss, oce_loss, regularization_loss)``; the loss is a SUM over pairs) and
for ``UNetModel.select_and_add_coordinates`` (``cellulus/models/unet.py:108-124``).
"""

import torch
import torch.nn as nn

from .. import _clx


def _grid(outputs):
    """(B, ND, Z, Y, X) extents of a planar (B, ND, [Z,] Y, X) tensor."""
    nd = outputs.ndim - 2
    if nd == 2:
        return outputs.shape[0], 2, 1, outputs.shape[2], outputs.shape[3]
    if nd == 3:
        return outputs.shape[0], 3, outputs.shape[2], outputs.shape[3], outputs.shape[4]
    raise ValueError(f"outputs must be (B, C, H, W) or (B, C, D, H, W), got {tuple(outputs.shape)}")


def _check_coords(outputs, coordinates):
    B, ND, Z, Y, X = _grid(outputs)
    if outputs.shape[1] != ND:
        raise ValueError("outputs must have one channel per spatial dimension")
    if coordinates.ndim != 3 or coordinates.shape[0] != B or coordinates.shape[2] != ND:
        raise ValueError(f"coordinates must be (B={B}, P, {ND}), got {tuple(coordinates.shape)}")
    if coordinates.dtype != torch.int64:
        raise TypeError("coordinates must be int64")
    return B, ND, Z, Y, X


def raise_on_bad_coordinates(count, spatial):
    """The reference indexes ``outputs[b, :, (z,) y, x]`` (unet.py:113-118): torch raises
    IndexError for a coordinate outside [-n, n).  The kernels skip and count such rows."""
    if count:
        raise IndexError(
            f"{count} coordinate row(s) index outside the network output of extent {tuple(spatial)} "
            "(coordinate column 0 is x = the LAST axis; the dataset's output_shape must equal the "
            "model's output extent)")


class _GatherAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, coordinates):
        _clx.require_device(outputs, "outputs")
        _clx.require_device(coordinates, "coordinates")
        B, ND, Z, Y, X = _check_coords(outputs, coordinates)
        outputs = outputs.contiguous()
        coordinates = coordinates.contiguous()
        P = coordinates.shape[1]
        sel = torch.empty((B, P, ND), dtype=torch.float32, device=outputs.device)
        oob = torch.empty(1, dtype=torch.int32, device=outputs.device)
        _clx.zero_many(oob)
        _clx.call("clx_gather_add_fwd", _clx.ptr(outputs), _clx.ptr(coordinates), _clx.ptr(sel),
                  B, P, ND, Z, Y, X, _clx.ptr(oob), _clx.stream_ptr(outputs.device))
        raise_on_bad_coordinates(int(oob.item()), (Z, Y, X)[3 - ND:])
        ctx.save_for_backward(coordinates)
        ctx.shape = tuple(outputs.shape)
        return sel

    @staticmethod
    def backward(ctx, dsel):
        (coordinates,) = ctx.saved_tensors
        shape = ctx.shape
        nd = len(shape) - 2
        B, ND = shape[0], shape[1]
        Z, Y, X = (1, shape[2], shape[3]) if nd == 2 else shape[2:]
        dsel = dsel.contiguous()
        dout = torch.empty(shape, dtype=torch.float32, device=dsel.device)
        _clx.zero_many(dout)
        _clx.call("clx_gather_add_bwd", _clx.ptr(dsel), _clx.ptr(coordinates), _clx.ptr(dout),
                  B, coordinates.shape[1], ND, Z, Y, X, None, _clx.stream_ptr(dsel.device))
        return dout, None


def gather_add(outputs, coordinates):
    return _GatherAdd.apply(outputs, coordinates)


class _OCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, reference, temperature, reg_weight):
        _clx.require_device(anchor, "anchor_embedding")
        _clx.require_device(reference, "reference_embedding")
        if anchor.shape != reference.shape:
            raise ValueError("anchor and reference embeddings must have the same shape")
        nd = anchor.shape[-1]
        a = anchor.contiguous()
        r = reference.contiguous()
        npairs = a.numel() // nd
        sums = torch.empty(3, dtype=torch.float64, device=a.device)
        _clx.zero_many(sums)
        need_grad = anchor.requires_grad
        da = torch.empty_like(a) if need_grad else None
        _clx.call("clx_oce_loss_fwd_bwd", _clx.ptr(a), _clx.ptr(r), _clx.ptr(da), _clx.ptr(sums),
                  npairs, nd, float(temperature), float(reg_weight), 1.0,
                  _clx.stream_ptr(a.device))
        ctx.da = da
        loss, oce, reg = sums.to(torch.float32).unbind(0)
        ctx.mark_non_differentiable(oce, reg)
        return loss, oce, reg

    @staticmethod
    def backward(ctx, g_loss, g_oce, g_reg):
        # d(loss)/da was produced in the forward launch; oce/reg are returned for logging only
        # (the reference never back-propagates through them separately).
        return ctx.da * g_loss, None, None, None


class OCELoss(nn.Module):  # type: ignore
    def __init__(
        self,
        temperature: float,
        regularization_weight: float,
        density: float,
        num_spatial_dims: int,
        device: torch.device,
    ):
        """Class definition for loss (same arguments as the reference, oce_loss.py:6-43)."""
        super().__init__()
        self.temperature = temperature
        self.regularization_weight = regularization_weight
        self.density = density
        self.num_spatial_dims = num_spatial_dims
        self.device = device

    def forward(self, anchor_embedding, reference_embedding):
        loss, oce_loss, regularization_loss = _OCE.apply(
            anchor_embedding, reference_embedding.detach(), self.temperature,
            self.regularization_weight)
        return loss, oce_loss, regularization_loss
