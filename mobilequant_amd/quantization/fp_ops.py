"""Floating-point leaf modules the surgery rules key on.

MobileQuant wraps plain tensor ops in modules so that hooks and module replacement can see them
(reference: mobilellm/model/ops.py:6-70, hf_model.py:162-201).  These are the minimal equivalents:
same class names and forward semantics, nothing else from the model zoo.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class FMatMul(nn.Module):
    """matmul as a module: attention's qk_bmm / pv_bmm (reference: ops.py:28-34)."""

    def forward(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return torch.matmul(a, b)


class ElementwiseAdd(nn.Module):
    def forward(self, x, y):
        return x + y


class ElementwiseMul(nn.Module):
    def forward(self, x, y):
        return x * y


class L2Norm(nn.Module):
    def __init__(self, p=2, dim=-1, eps=1e-12):
        super().__init__()
        self.p, self.dim, self.eps = p, dim, eps

    def forward(self, x):
        return F.normalize(x, p=self.p, dim=self.dim, eps=self.eps)


class HFRMSNorm(nn.Module):
    """RMSNorm with the two evaluation modes of the reference (hf_model.py:162-201):
    ``x * rsqrt(mean(x^2) + eps)`` in fp32, or ``sqrt(dim) * l2normalize(x)`` when
    ``l2norm_as_rmsnorm`` (the form the NPU export uses); then ``weight * y (+ bias)``."""

    def __init__(self, dim: int, eps: float = 1e-6, device=None, dtype=None, bias=None, l2norm_as_rmsnorm=False):
        super().__init__()
        self.eps = eps
        self.alpha = math.sqrt(dim)
        self.weight = nn.Parameter(torch.empty(dim, device=device, dtype=dtype))
        self.bias = None if bias is None else nn.Parameter(torch.zeros(dim, device=device, dtype=dtype))
        self.l2norm_as_rmsnorm = l2norm_as_rmsnorm
        if l2norm_as_rmsnorm:
            self.l2norm = L2Norm()
        self.elementwisemul = ElementwiseMul()
        nn.init.normal_(self.weight)

    @property
    def variance_epsilon(self):
        return self.eps

    def forward_impl(self, x, weight, bias):
        if self.l2norm_as_rmsnorm:
            y = self.alpha * self.l2norm(x)
        else:
            xf = x.float()
            y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).type_as(x)
        y = self.elementwisemul(weight, y)
        if bias is not None:
            y = y + bias
        return y

    def forward(self, x):
        return self.forward_impl(x, self.weight, self.bias)
