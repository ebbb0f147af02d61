"""Shared builders for the parity tests (synthetic scenes of SURVEY 8d, scaled down)."""
import ctypes

import numpy as np
import torch

from dreammat_amd import envlight as penv
from dreammat_amd import mesh as pmesh
from oracle import camera, envlight as oenv, field as ofield, raster as oraster


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def make_views(B, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    elev = torch.rand(B, generator=g) * 65 - 20
    azim = (torch.rand(B, generator=g) + torch.arange(B)) / B * 360 - 180
    dist = torch.rand(B, generator=g) + 3.0
    fovy = torch.rand(B, generator=g) * 20 + 25
    return camera.camera_batch(elev, azim, dist, fovy, H, W)


def synthetic_latlong(seed, h=64, w=128):
    """seeded log-normal sky + one sun lobe."""
    g = torch.Generator().manual_seed(1000 + seed)
    sky = torch.exp(torch.randn(h // 8, w // 8, 3, generator=g))
    sky = torch.nn.functional.interpolate(sky.permute(2, 0, 1)[None], (h, w), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    d = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0)
    v = torch.linspace(0, np.pi, h)[:, None].expand(h, w)
    u = torch.linspace(-np.pi, np.pi, w)[None, :].expand(h, w)
    dirs = torch.stack([torch.sin(v) * torch.sin(u), torch.cos(v), -torch.sin(v) * torch.cos(u)], -1)
    lobe = 50.0 * torch.exp(-(1 - (dirs * d).sum(-1)) / 0.01)
    return (sky * 0.5 + lobe[..., None]).contiguous()


def mesh_dict(m):
    tri = m.t_pos_idx.numpy().astype(np.int32)
    return dict(v_pos=m.v_pos.numpy(), v_nrm=m.v_nrm.numpy(), t_pos_idx=tri, opp=oraster.build_topology(tri))
