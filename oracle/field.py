"""Feature field of DreamMatMesh.forward (TEST INFRASTRUCTURE; fp32 torch, autograd-friendly).

Follows threestudio/models/geometry/dreammat_mesh.py:239-254 (forward), geometry/base.py:20-32
(contract_to_unisphere, bounded branch: (x - bbox0)/(bbox1-bbox0), bbox = +-radius),
networks.py:55-64 (TCNNEncoding, fp32 output) and networks.py:150-187 (VanillaMLP: bias-free
Linear(32,64) -> ReLU -> Linear(64,5), no output activation).

tiny-cuda-nn (requirements.txt:6, unpinned) is not in the tree; its multiresolution HashGrid is
restated from its published algorithm (PARITY UNPINNED):
  scale_l = base_res * per_level_scale^l - 1 ; res_l = ceil(scale_l) + 1
  params_l = min(next_multiple(res_l^3, 8), 2^log2_hashmap_size)      (x n_features)
  pos = x*scale_l + 0.5 ; cell = floor(pos) ; w = pos - cell ; trilinear over the 8 corners
  index = dense (x + y*res + z*res^2) while the running stride stays <= params_l, otherwise
          (x*1) ^ (y*2654435761) ^ (z*805459861)  (uint32); finally  index % params_l
  output = concat over levels of n_features values.
"""
import math

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)


def grid_levels(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16,
                per_level_scale=1.447269237440378, n_dims=3):
    """n_dims = 2: the uv-space field (dreammat_mesh.py:128-135, n_input_dims = 2): the same rules with res^2 entries per
    dense level and two corners per axis"""
    levels = []
    offset = 0
    for l in range(n_levels):
        scale = np.float32(math.pow(2.0, l * math.log2(per_level_scale)) * base_resolution - 1.0)
        res = int(math.ceil(float(scale))) + 1
        n = res ** n_dims
        n = (n + 7) // 8 * 8
        n = min(n, 1 << log2_hashmap_size)
        levels.append({"scale": float(scale), "res": res, "size": n, "offset": offset})
        offset += n
    return levels, offset  # offset = total entries (x n_features params)


def hash_encode(x, table, levels, n_features=2):
    """x [N,3] (or [N,2]: uv-space field) in [0,1] (unclamped), table [total, F] -> [N, L*F]."""
    D = x.shape[1]
    outs = []
    for lv in levels:
        scale, res, size, off = lv["scale"], lv["res"], lv["size"], lv["offset"]
        pos = x * scale + 0.5
        cell = pos.floor()
        w = pos - cell
        cell = cell.long()
        acc = 0
        for corner in range(1 << D):
            idx_parts = []
            wgt = 1.0
            for d in range(D):
                bit = (corner >> d) & 1
                idx_parts.append(cell[:, d] + bit)
                wgt = wgt * (w[:, d] if bit else (1 - w[:, d]))
            # dense index while stride <= size
            stride = 1
            index = torch.zeros_like(idx_parts[0])
            for d in range(D):
                if stride <= size:
                    index = index + (idx_parts[d] & 0xFFFFFFFF) * stride
                    stride *= res
            index = index & 0xFFFFFFFF
            if size < stride:
                h = torch.zeros_like(index)
                for d in range(D):
                    h = h ^ (((idx_parts[d] & 0xFFFFFFFF) * PRIMES[d]) & 0xFFFFFFFF)
                index = h
            index = index % size
            acc = acc + table[off + index] * wgt[:, None]
        outs.append(acc)
    return torch.cat(outs, dim=-1)


def contract_to_unisphere(x, radius=1.0):
    return (x + radius) / (2 * radius)


def field_forward(points, table, w1, w2, levels, radius=1.0, return_hidden=False):
    """points [N,3] world (or [N,2] texture coordinates, contracted with the same +-radius box: base.py:209-219 bbox2d) ->
    features [N,5].  w1 [64,32], w2 [5,64] (nn.Linear layout, no bias).
    return_hidden: also the pre-activation of the hidden layer [N,64] (tests use it to find rows that sit on a ReLU kink,
    where two correct fp32 evaluations route the gradient differently)."""
    enc = hash_encode(contract_to_unisphere(points, radius), table, levels)
    pre = enc @ w1.t()
    out = torch.relu(pre) @ w2.t()
    return (out, pre.detach()) if return_hidden else out
