"""`dreammat-mesh` geometry plugin (threestudio/models/geometry/dreammat_mesh.py:89-274).

Fixed triangle mesh + learnable feature field: multiresolution hash grid (HIP kernel, replaces
tcnn.Encoding -- threestudio/models/networks.py:55-64) followed by the bias-free VanillaMLP
32 -> 64 (ReLU) -> n_feature_dims (networks.py:150-187; a plain GEMM, left to hipBLASLt).
state_dict keys match the reference: `encoding.encoding.params`, `feature_network.layers.{0,2}.weight`.
The never-used predictor heads of the reference (dreammat_mesh.py:136-139) are NOT instantiated:
they carry no gradient and would only bloat the all-reduce buffer.
"""
import os
from dataclasses import dataclass, field
from typing import Any, Optional

import torch
import torch.nn as nn

import dreammat_amd
from . import hipops, mesh as meshlib
from .base import BaseModule


class HashGridEncoding(nn.Module):
    """Drop-in for TCNNEncoding (networks.py:55-64): `.n_output_dims`, `.encoding.params` flat fp32."""

    class _Inner(nn.Module):
        def __init__(self, n_params):
            super().__init__()
            # tcnn initialises hash-grid params U(-1e-4, 1e-4)
            self.params = nn.Parameter((torch.rand(n_params) * 2 - 1) * 1e-4)

    def __init__(self, in_channels, config, radius=1.0):
        super().__init__()
        assert in_channels in (2, 3) and config.get("otype", "HashGrid") == "HashGrid", \
            "only the HashGrid encoding of dreammat.yaml:43-49 is implemented (3-D points, or 2-D texture coordinates)"
        self.spec = hipops.GridSpec(config["n_levels"], config["n_features_per_level"], config["log2_hashmap_size"],
                                    config["base_resolution"], config["per_level_scale"], n_dims=in_channels)
        self.n_input_dims = in_channels
        self.n_output_dims = self.spec.n_output_dims
        self.radius = radius
        self.encoding = self._Inner(self.spec.n_params)

    def forward(self, x_world):
        """x_world [M,3] in world space (or [M,2] texture coordinates); contract_to_unisphere is fused into the kernel."""
        p = self.encoding.params
        # when the trainer has re-homed the parameters into its flat buffer (system.FlatParams), the
        # backward kernel accumulates straight into that gradient slice
        sink = p.grad if (p.grad is not None and torch.is_grad_enabled() and p.requires_grad) else None
        return hipops.hashgrid_encode(x_world, p, self.spec, self.radius, grad_sink=sink)


def _wgrad_splitk(g, h, n_split=512):
    """dW = g @ h^T for feature-major g [out, N], h [in, N] with N in the millions: one GEMM with K = N runs as a
    single skinny tile (2.2 ms for 64x32xN=2.6M on MI355X); split the points into `n_split` slabs, one batched
    GEMM over strided views (no copies), and add the slab results."""
    N = g.shape[1]
    n = N // n_split
    if n < 64:
        return g @ h.t()
    g = g.contiguous()
    h = h.contiguous()
    Np = n * n_split
    gp = g[:, :Np].view(g.shape[0], n_split, n).transpose(0, 1)          # [S, out, n]  (row stride N)
    hp = h[:, :Np].view(h.shape[0], n_split, n).permute(1, 2, 0)         # [S, n, in]   (column stride N)
    dW = torch.bmm(gp, hp).sum(0)
    if Np < N:
        dW = dW + g[:, Np:] @ h[:, Np:].t()
    return dW


class _FeatureMajorLinear(torch.autograd.Function):
    """y = W @ h for feature-major activations h [in, N]."""

    @staticmethod
    def forward(ctx, W, h):
        ctx.save_for_backward(W, h)
        return W @ h

    @staticmethod
    def backward(ctx, g):
        W, h = ctx.saved_tensors
        dW = _wgrad_splitk(g, h) if ctx.needs_input_grad[0] else None
        dh = W.t() @ g if ctx.needs_input_grad[1] else None
        return dW, dh


class VanillaMLP(nn.Module):
    def __init__(self, dim_in, dim_out, config):
        super().__init__()
        n_neurons, n_hidden = config["n_neurons"], config["n_hidden_layers"]
        layers = [nn.Linear(dim_in, n_neurons, bias=False), nn.ReLU(inplace=True)]
        for _ in range(n_hidden - 1):
            layers += [nn.Linear(n_neurons, n_neurons, bias=False), nn.ReLU(inplace=True)]
        layers += [nn.Linear(n_neurons, dim_out, bias=False)]
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        x = x.float()
        if x.dim() == 2 and x.stride(0) == 1 and x.shape[0] > 1:
            # feature-major input (the hash-grid kernel's coalesced layout): keep the whole MLP
            # feature-major (W @ X^T) and hand back an [M, out] view -- same numbers, no transposes
            h = x.t()
            if len(self.layers) == 3 and hipops.field_mlp_ok(h, self.layers[0].weight, self.layers[2].weight):
                # the shipped shape (2L -> 64 -> ReLU -> n_out): both layers in one kernel, the hidden layer in registers
                return hipops.field_mlp(h, self.layers[0].weight, self.layers[2].weight).t()
            for layer in self.layers:
                h = _FeatureMajorLinear.apply(layer.weight, h) if isinstance(layer, nn.Linear) else torch.relu(h)
            return h.t()
        return self.layers(x)


@dreammat_amd.register("dreammat-mesh")
class DreamMatMesh(BaseModule):
    @dataclass
    class Config(BaseModule.Config):
        radius: float = 1.0
        n_input_dims: int = 3
        n_feature_dims: int = 5
        pos_encoding_config: dict = field(default_factory=lambda: {
            "otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
            "base_resolution": 16, "per_level_scale": 1.447269237440378})
        mlp_network_config: dict = field(default_factory=lambda: {
            "otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": 64,
            "n_hidden_layers": 1})
        shape_init: str = ""
        shape_init_params: Optional[Any] = None
        shape_init_mesh_up: str = "+z"
        shape_init_mesh_front: str = "+x"

    cfg: Config

    def configure(self) -> None:
        if self.cfg.n_input_dims not in (2, 3):
            raise ValueError(f"n_input_dims={self.cfg.n_input_dims}: 3 (surface points) or 2 (texture coordinates)")
        # n_input_dims = 2 (dreammat_mesh.py:128-135, 246-250): the field lives in texture space -- the renderer queries it at
        # the interpolated uv of a pixel, contracted with bbox2d = +-radius like the 3-D points (geometry/base.py:209-219)
        self.encoding = HashGridEncoding(self.cfg.n_input_dims, self.cfg.pos_encoding_config, radius=self.cfg.radius)
        self.feature_network = VanillaMLP(self.encoding.n_output_dims, self.cfg.n_feature_dims,
                                          self.cfg.mlp_network_config)
        self.mesh = self._load_mesh()

    def _load_mesh(self):
        si = self.cfg.shape_init
        if si.startswith("mesh:"):
            m = meshlib.load_obj(si[5:])
            # axis alignment (dreammat_mesh.py:175-199): rotate so that cfg up/front map to +z/+x
            dirs = {"+x": [1, 0, 0], "+y": [0, 1, 0], "+z": [0, 0, 1], "-x": [-1, 0, 0], "-y": [0, -1, 0], "-z": [0, 0, -1]}
            z_ = torch.tensor(dirs[self.cfg.shape_init_mesh_up], dtype=torch.float32)
            x_ = torch.tensor(dirs[self.cfg.shape_init_mesh_front], dtype=torch.float32)
            y_ = torch.cross(z_, x_, dim=0)
            R = torch.stack([x_, y_, z_], dim=0)     # rows: new axes in old coordinates
            m.v_pos = (m.v_pos @ R.t()).contiguous()
            if m._v_nrm is not None:
                m._v_nrm = (m._v_nrm @ R.t()).contiguous()
            meshlib.normalize_mesh(m, float(self.cfg.shape_init_params))
        elif si == "quad":
            m = meshlib.quad_mesh()
        elif si.startswith("sphere"):
            parts = si.split(":")
            n_lon, n_lat = (int(parts[1]), int(parts[2])) if len(parts) == 3 else (160, 160)
            m = meshlib.displaced_sphere(n_lon, n_lat)
        else:
            raise ValueError(f"Unknown shape initialization type: {si}")
        m.t_pos_idx = m.t_pos_idx.to(torch.int32).contiguous()
        _ = m.v_nrm
        self.register_buffer("v_buffer", m.v_pos)
        self.register_buffer("t_buffer", m.t_pos_idx)
        self.register_buffer("vnrm_buffer", m.v_nrm)
        if m.v_tex is not None:       # dreammat_mesh.py:211-214: the exporter rasterizes in UV space on the device
            self.register_buffer("vtex_buffer", m.v_tex.contiguous())
        return m

    def isosurface(self):
        # buffers follow .to(device); rebuild the view lazily (dreammat_mesh.py:230-237)
        m = meshlib.Mesh(self.v_buffer, self.t_buffer, self.vnrm_buffer, getattr(self, "vtex_buffer", None))
        return m

    def forward(self, points, output_normal: bool = False):
        assert not output_normal, f"Normal output is not supported for {self.__class__.__name__}"
        enc = self.encoding(points.reshape(-1, self.cfg.n_input_dims))
        features = self.feature_network(enc).view(*points.shape[:-1], self.cfg.n_feature_dims)
        return {"features": features}

    def export(self, points, **kwargs):
        return self.forward(points)
