"""Free-function geometry API of the reference (``protstruc.geometry``), on the GPU.

``angle``, ``dihedral`` and ``gram_schmidt`` evaluate in the same HIP device
functions the batch kernels use (csrc/ps_common.hpp) through a point-wise
launcher; ``dot`` / ``norm`` / ``unit`` are single broadcasting tensor ops.
Type polymorphism follows the reference's ``with_tensor`` decorator
(decorator.py:5-53): numpy arrays in -> numpy arrays out (float64 is computed in
float32, as there), any tensor in -> tensor out.  Tensors must live on (or are
moved to) the GPU; there is no CPU evaluation path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


# ideal backbone geometry (reference constants/ideal.py:2-36): bond lengths in Angstrom, angles in radians
IDEAL_NA, IDEAL_AC, IDEAL_NAC = 1.458, 1.523, 1.937


def ideal_backbone_coordinates(size, include_cb: bool = False) -> torch.Tensor:
    """Ideal N, CA, C (and CB) coordinates with CA at the origin and CA->C along +x, expanded to
    ``(*size, 3 or 4, 3)`` (reference geometry.py:191-226).  Returned on the CPU like the reference."""
    import math

    ca = torch.zeros(3)
    c = torch.tensor([IDEAL_AC, 0.0, 0.0])
    n = torch.tensor([IDEAL_NA * math.cos(IDEAL_NAC), IDEAL_NA * math.sin(IDEAL_NAC), 0.0])
    atoms = [n, ca, c]
    if include_cb:
        b_, c_ = ca - n, c - ca
        a_ = torch.linalg.cross(b_, c_)
        atoms.append(-0.58273431 * a_ + 0.56802827 * b_ - 0.54067466 * c_ + ca)  # tetrahedral CB placement
    return torch.stack(atoms).expand(*size, -1, -1)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("protstruc_amd.geometry is HIP-only (no CPU fallback): no GPU is visible")
    return torch.device("cuda", torch.cuda.current_device())


def _prep(args):
    """with_tensor semantics: convert ndarrays, remember whether any input was already a tensor."""
    found_tensor = any(isinstance(a, torch.Tensor) for a in args)
    dev = next((a.device for a in args if isinstance(a, torch.Tensor) and a.is_cuda), None) or _device()
    out = []
    for a in args:
        if isinstance(a, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(a))
            a = t.float() if t.dtype in (torch.float32, torch.float64) else t
        out.append(a.to(dev))
    return out, found_tensor


def _finish(t, found_tensor):
    return t if found_tensor else t.cpu().numpy()


def dot(x, y):
    """(x*y).sum(-1, keepdim=True)  (reference geometry.py:24-26)."""
    (x, y), ft = _prep([x, y])
    return _finish((x * y).sum(dim=-1, keepdim=True), ft)


def norm(x):
    """x.norm(dim=-1, keepdim=True)  (reference geometry.py:29-31)."""
    (x,), ft = _prep([x])
    return _finish(x.norm(dim=-1, keepdim=True), ft)


def unit(x):
    """x / |x|  (reference geometry.py:34-36)."""
    (x,), ft = _prep([x])
    return _finish(x / x.norm(dim=-1, keepdim=True), ft)


def angle(a, b, c, to_degree: bool = False):
    """Planar angle a-b-c in [0, pi] (reference geometry.py:39-71); acos without clamp, as in the reference."""
    (a, b, c), ft = _prep([a, b, c])
    out = ops.pointwise(0, a, b, c)
    return _finish(torch.rad2deg(out) if to_degree else out, ft)


def dihedral(a, b, c, d, to_degree: bool = False):
    """Dihedral angle of a-b-c-d in [-pi, pi] (reference geometry.py:74-124)."""
    (a, b, c, d), ft = _prep([a, b, c, d])
    out = ops.pointwise(1, a, b, c, d)
    return _finish(torch.rad2deg(out) if to_degree else out, ft)


def gram_schmidt(a, b, c):
    """Orthonormal basis of the plane through (c-b) and (a-b), basis vectors as columns (reference geometry.py:413-439)."""
    (a, b, c), ft = _prep([a, b, c])
    return _finish(ops.pointwise(2, a, b, c), ft)


def kabsch(a, b):
    """Rotation (3,3) and translation (3,) minimising the RMSD of ``R a + t`` against ``b`` for point sets
    (n,3) (reference geometry.py:442-480); evaluated by the batched Kabsch kernel with a batch of one."""
    (a, b), ft = _prep([a, b])
    mask = torch.ones(1, a.shape[0], dtype=torch.bool, device=a.device)
    R, t = ops.kabsch(a.reshape(1, -1, 1, 3).contiguous(), b.reshape(1, -1, 1, 3).contiguous(), mask)
    return _finish(R[0], ft), _finish(t[0], ft)
