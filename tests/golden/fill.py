"""Deterministic, name-keyed parameter fill shared by the golden generator (which drives the
REFERENCE model) and by the tests (which drive sigma_amd's model and the oracle model).

Both sides build their model, then call ``fill_parameters(model)``: every parameter is
overwritten by a function of (its state_dict name, its shape) only, so no weights have to be
stored.  Values are non-trivial for EVERY parameter (so each one influences the logits) but
keep activations O(1): fan-in scaled matrices, LayerNorm scales around 1, A = -exp(A_log)
around -(1..N), dt biases = softplus^-1 of log-uniform[1e-3, 1e-1] as in the model's own init.
"""
import math
import zlib

import torch


def _gen(name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return g


def value_for(name: str, shape) -> torch.Tensor:
    g = _gen(name)
    leaf = name.split(".")[-1]
    shape = tuple(shape)
    if leaf in ("A_logs", "A_log_1", "A_log_2"):
        n = shape[-1]
        base = torch.log(torch.arange(1, n + 1, dtype=torch.float32)).expand(shape)
        return base + 0.1 * torch.randn(shape, generator=g)
    if leaf in ("Ds", "D_1", "D_2", "scale1", "scale2"):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == "dt_projs_bias" or (leaf == "bias" and ".dt_proj_" in name):
        dt = torch.exp(torch.rand(shape, generator=g) * (math.log(0.1) - math.log(0.001)) + math.log(0.001))
        return dt + torch.log(-torch.expm1(-dt))
    if len(shape) == 1:
        if leaf == "weight":                       # LayerNorm scale
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)   # biases
    if leaf == "dt_projs_weight" or (leaf == "weight" and ".dt_proj_" in name):
        r = shape[-1]
        return (torch.rand(shape, generator=g) * 2 - 1) * r ** -0.5
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    if leaf == "x_proj_weight":
        fan_in = shape[-1]
    return torch.randn(shape, generator=g) * fan_in ** -0.5


@torch.no_grad()
def fill_parameters(model: torch.nn.Module) -> None:
    for name, p in model.state_dict().items():
        p.copy_(value_for(name, p.shape).to(p.dtype))


def make_inputs(batch: int, height: int, width: int, num_classes: int, seed: int = 0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    rgb = torch.randn(batch, 3, height, width, generator=g)
    x = torch.randn(batch, 3, height, width, generator=g)
    label = torch.randint(0, num_classes, (batch, height, width), generator=g)
    label[:, :2, :] = 255                           # exercise ignore_index
    return rgb, x, label
