"""torch-facing wrappers of the C ABI (include/dreammat_hip.h): device pointers + the current HIP
stream go straight into libdreammat_hip.so; `torch.autograd.Function`s provide the backward passes.

PyTorch is plumbing here (memory, streams, autograd graph) -- all arithmetic of these ops runs in the
hand-written gfx950 kernels.  There is no fallback: CPU tensors or a missing library raise.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import check


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- optional per-kernel HIP-event timing (bench.py's live roofline numbers) --------------------
_TIMING = {"on": False, "events": {}, "only": None}


def enable_kernel_timing(on=True, only=None):
    """HIP events around every launch below (`only`: key prefixes to restrict it to -- an event pair costs ~2 us of
    stream time, so a measured region should time only the kernels it reports)."""
    _TIMING["on"] = on
    _TIMING["only"] = tuple(only) if only else None
    if on:
        _TIMING["events"] = {}


class _Timed:
    """Brackets one launch with HIP events on the stream it is enqueued on (torch's current stream)."""

    def __init__(self, key, work=0.0):
        self.key, self.work = key, work

    def __enter__(self):
        self.live = _TIMING["on"] and (_TIMING["only"] is None or self.key.startswith(_TIMING["only"]))
        if self.live:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.live:
            self.e1.record()
            _TIMING["events"].setdefault(self.key, []).append((self.e0, self.e1, self.work))


def kernel_times():
    """{key: {"launches", "avg_ms", "work_per_launch"}} -- call after torch.cuda.synchronize()."""
    out = {}
    for k, evs in _TIMING["events"].items():
        ms = [a.elapsed_time(b) for a, b, _ in evs]
        out[k] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "work_per_launch": sum(w for _, _, w in evs) / len(evs)}
    return out


HALF_DTYPES = (torch.bfloat16, torch.float16)     # element types the net kernels are built for (csrc/dm_elem.h)


def _sym(name, dtype):
    """entry point of `name` (a bf16 name of include/dreammat_hip.h) for `dtype`: the IEEE-half instantiation of a net kernel
    carries f16 in place of bf16, or an _f16 suffix where the name has no dtype token -> (callable, symbol name)"""
    if dtype == torch.float16:
        name = name.replace("bf16", "f16") if "bf16" in name else name + "_f16"
    elif dtype != torch.bfloat16:
        raise TypeError(f"{name}: bf16 or f16 tensors expected, got {dtype}")
    return getattr(_lib.lib(), name), name


def _same_half(*ts):
    dt = ts[0].dtype
    assert dt in HALF_DTYPES and all(t is None or t.dtype == dt for t in ts), [None if t is None else t.dtype for t in ts]
    return dt


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DmError("dreammat_amd HIP ops need tensors on the MI355X (got a CPU tensor); "
                               "there is no CPU fallback")


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _rs_cs(t):
    """(row stride, col stride) in elements of a 2-D [N,C] view."""
    return int(t.stride(0)), int(t.stride(1))


# ------------------------------------------------------------------------------------------ mesh
def build_topology(tri):
    """[Nf,3] int tensor (any device) -> opp [Nf,3] int32 on the same device."""
    t = tri.detach().to("cpu", torch.int32).contiguous().numpy()
    opp = np.empty_like(t)
    check(_lib.lib().dm_mesh_build_topology(t.ctypes.data, t.shape[0], opp.ctypes.data), "dm_mesh_build_topology")
    return torch.from_numpy(opp).to(tri.device)


# ------------------------------------------------------------------------------------------ raster
def vertex_transform(v_pos, mvp):
    _need_cuda(v_pos, mvp)
    v_pos, mvp = _f32c(v_pos), _f32c(mvp)
    B, Nv = mvp.shape[0], v_pos.shape[0]
    out = torch.empty(B, Nv, 4, device=v_pos.device, dtype=torch.float32)
    check(_lib.lib().dm_vertex_transform(v_pos.data_ptr(), Nv, mvp.data_ptr(), B, out.data_ptr(), _stream()),
          "dm_vertex_transform")
    return out


class RasterContext:
    """Plays the role of dr.RasterizeCudaContext (threestudio/utils/rasterize.py:12-20): owns the
    triangle-bin workspace, reused across steps."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.ws = None
        self.ws_mult = 1

    def _workspace(self, B, Nf, H, W):
        need = int(_lib.lib().dm_raster_workspace_bytes(B, Nf, H, W)) * self.ws_mult
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.ws

    def rasterize(self, pos_clip, tri, H, W, check_overflow=False):
        _need_cuda(pos_clip, tri)
        pos_clip = _f32c(pos_clip)
        assert tri.dtype == torch.int32 and tri.is_contiguous()
        B, Nv = pos_clip.shape[0], pos_clip.shape[1]
        rast = torch.empty(B, H, W, 4, device=pos_clip.device, dtype=torch.float32)
        while True:
            ws = self._workspace(B, tri.shape[0], H, W)
            # read pos B Nv 16 + tri Nf 12, write rast P 16 (reported, not gated: binning / latency bound)
            with _Timed("rasterize", 16.0 * B * Nv + 12.0 * tri.shape[0] + 16.0 * B * H * W):
                check(_lib.lib().dm_rasterize(pos_clip.data_ptr(), B, Nv, tri.data_ptr(), tri.shape[0], H, W,
                                              rast.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "dm_rasterize")
            if not check_overflow:
                return rast
            flag = ctypes.c_int(0)
            check(_lib.lib().dm_raster_overflowed(ws.data_ptr(), _stream(), ctypes.byref(flag)), "dm_raster_overflowed")
            if not flag.value:
                return rast
            self.ws_mult *= 2
            self.ws = None


def interpolate(attr, rast, tri):
    _need_cuda(attr, rast, tri)
    attr, rast = _f32c(attr), _f32c(rast)
    C = attr.shape[-1]
    out = torch.empty(*rast.shape[:-1], C, device=rast.device, dtype=torch.float32)
    npix = rast.numel() // 4
    check(_lib.lib().dm_interpolate(attr.data_ptr(), attr.shape[0], C, tri.data_ptr(), rast.data_ptr(), npix,
                                    out.data_ptr(), _stream()), "dm_interpolate")
    return out


def antialias_plan(pos_clip, tri, opp, rast):
    _need_cuda(pos_clip, tri, opp, rast)
    B, H, W, _ = rast.shape
    plan = torch.empty(B, H, W, 2, device=rast.device, dtype=torch.float32)
    with _Timed("antialias_plan", 24.0 * B * H * W):                          # read rast 16, write plan 8 per pixel
        check(_lib.lib().dm_antialias_plan(pos_clip.data_ptr(), B, pos_clip.shape[1], tri.data_ptr(), opp.data_ptr(),
                                           rast.data_ptr(), H, W, plan.data_ptr(), _stream()), "dm_antialias_plan")
    return plan


class _Antialias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, plan):
        _need_cuda(color, plan)
        color = _f32c(color)
        B, H, W, C = color.shape
        out = torch.empty_like(color)
        # read colour + rast P (4 C + 16), write P 4 C (the kernel reads the 8-byte plan instead of rast)
        with _Timed(f"antialias_apply[C={C}]", float(B * H * W) * (8 * C + 16)):
            check(_lib.lib().dm_antialias_apply(color.data_ptr(), plan.data_ptr(), B, H, W, C, out.data_ptr(), _stream()),
                  "dm_antialias_apply")
        ctx.save_for_backward(plan)
        return out

    @staticmethod
    def backward(ctx, g):
        (plan,) = ctx.saved_tensors
        g = _f32c(g)
        B, H, W, C = g.shape
        dcolor = torch.empty_like(g)
        with _Timed(f"antialias_grad[C={C}]", float(B * H * W) * (8 * C + 16)):
            check(_lib.lib().dm_antialias_grad(g.data_ptr(), plan.data_ptr(), B, H, W, C, dcolor.data_ptr(), _stream()),
                  "dm_antialias_grad")
        return dcolor, None


def antialias(color, plan):
    return _Antialias.apply(color, plan)


class GBuffer:
    """Compacted covered-pixel G-buffer (SoA, pitch = cap).  `n` is the host copy of the row count."""

    def __init__(self, pix_idx, pos, pos_jitter, nrm, view, n_dev, n):
        self.pix_idx, self.pos, self.pos_jitter, self.nrm, self.view = pix_idx, pos, pos_jitter, nrm, view
        self.n_dev, self.n = n_dev, n


def gbuffer_order():
    """Enumeration order of the compacted G-buffer rows: "tile" (default; 8 x 8 pixel blocks per wave, csrc/raster.hip) or
    "row" (the reference's row-major `x[selector]` order) -- DREAMMAT_GBUF_ORDER, a measurement knob: every consumer goes
    through pix_idx, so results do not depend on it."""
    o = os.environ.get("DREAMMAT_GBUF_ORDER", "tile")
    if o not in ("tile", "row"):
        raise ValueError(f"DREAMMAT_GBUF_ORDER={o!r}: expected 'tile' or 'row'")
    return o


def gbuffer_compact(rast, tri, v_pos, v_nrm, rays_d, jitter_u=None, jitter_n=None, jitter_eps=0.05, order=None):
    _need_cuda(rast, tri, v_pos, v_nrm, rays_d)
    dev = rast.device
    npix = rast.numel() // 4
    order = order or gbuffer_order()
    if order == "tile" and rast.dim() != 4:
        raise ValueError("tile-ordered G-buffer needs rast as [B,H,W,4]")
    cap = (npix + 3) // 4 * 4              # SoA rows 16-byte aligned (the shade kernels stream them 16 B per lane)
    rays_d = _f32c(rays_d)
    pix_idx = torch.empty(cap, dtype=torch.int32, device=dev)
    pos = torch.empty(3, cap, device=dev)
    nrm = torch.empty(3, cap, device=dev)
    view = torch.empty(3, cap, device=dev)
    pos_j = torch.empty(3, cap, device=dev) if jitter_u is not None else None
    n_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    L = _lib.lib()
    if order == "tile":
        B, H, W = rast.shape[:3]
        wsb = int(L.dm_gbuffer_tiled_workspace_bytes(B, H, W))
    else:
        wsb = int(L.dm_gbuffer_workspace_bytes(npix))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ju_t = _f32c(jitter_u) if jitter_u is not None else None   # keep the (possibly temporary) tensors alive
    jn_t = _f32c(jitter_n) if jitter_n is not None else None
    ju = ju_t.data_ptr() if ju_t is not None else None
    jn = jn_t.data_ptr() if jn_t is not None else None
    tail = (tri.data_ptr(), _f32c(v_pos).data_ptr(), _f32c(v_nrm).data_ptr(), rays_d.data_ptr(), ju, jn, float(jitter_eps), cap,
            pix_idx.data_ptr(), pos.data_ptr(), pos_j.data_ptr() if pos_j is not None else None, nrm.data_ptr(),
            view.data_ptr(), n_dev.data_ptr(), ws.data_ptr(), wsb, _stream())
    if order == "tile":
        check(L.dm_gbuffer_compact_tiled(rast.data_ptr(), B, H, W, *tail), "dm_gbuffer_compact_tiled")
    else:
        check(L.dm_gbuffer_compact(rast.data_ptr(), npix, *tail), "dm_gbuffer_compact")
    n = int(n_dev.item())   # the one host sync of the render path (sizes the torch-side tensors)
    return GBuffer(pix_idx[:n], pos[:, :n], pos_j[:, :n] if pos_j is not None else None, nrm[:, :n], view[:, :n],
                   n_dev, n)


def control_maps(rast, tri, v_nrm, w2c):
    _need_cuda(rast, tri, v_nrm, w2c)
    B, H, W, _ = rast.shape
    depth = torch.empty(B, H, W, 1, device=rast.device)
    normal = torch.empty(B, H, W, 3, device=rast.device)
    mm = torch.empty(2 * B, dtype=torch.int32, device=rast.device)
    check(_lib.lib().dm_control_maps(rast.data_ptr(), B, H, W, tri.data_ptr(), _f32c(v_nrm).data_ptr(),
                                     _f32c(w2c).data_ptr(), depth.data_ptr(), normal.data_ptr(), mm.data_ptr(),
                                     _stream()), "dm_control_maps")
    return depth, normal


class _ScatterRows(torch.autograd.Function):
    """dense[pix_idx] = rows  (raytracing_renderer.py:198); backward = gather."""

    @staticmethod
    def forward(ctx, rows, pix_idx, n_dev, dense_init):
        _need_cuda(rows, pix_idx, dense_init)
        C = rows.shape[1]
        out = dense_init.clone()
        if rows.shape[0] > 0:                        # nothing covered (all views empty): the init IS the result
            rs, cs = _rs_cs(rows)
            check(_lib.lib().dm_scatter_rows(pix_idx.data_ptr(), n_dev.data_ptr(), rows.shape[0], rows.data_ptr(),
                                             rs, cs, C, out.data_ptr(), _stream()), "dm_scatter_rows")
        ctx.save_for_backward(pix_idx, n_dev)
        ctx.n = rows.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        pix_idx, n_dev = ctx.saved_tensors
        g = _f32c(g)
        C = g.shape[-1]
        n_p = (ctx.n + 3) // 4 * 4                     # SoA with 16-byte aligned channels (the shade backward streams 16 B per lane)
        d = torch.empty(C, n_p, device=g.device)[:, :ctx.n]
        if ctx.n > 0:
            check(_lib.lib().dm_gather_rows(pix_idx.data_ptr(), n_dev.data_ptr(), ctx.n, g.data_ptr(), C,
                                            d.data_ptr(), 1, n_p, _stream()), "dm_gather_rows")
        return d.t(), None, None, None


def gather_rows(dense, pix_idx, n_dev, n):
    """dense [P,C] contiguous fp32 -> its rows at the covered pixels, [n,C] (no gradient: attributes of the fixed mesh)."""
    _need_cuda(dense, pix_idx, n_dev)
    dense = _f32c(dense)
    C = dense.shape[-1]
    out = torch.empty(n, C, device=dense.device, dtype=torch.float32)
    if n > 0:
        check(_lib.lib().dm_gather_rows(pix_idx.data_ptr(), n_dev.data_ptr(), n, dense.data_ptr(), C, out.data_ptr(), C, 1, _stream()),
              "dm_gather_rows")
    return out


def scatter_rows(rows, pix_idx, n_dev, dense_init):
    """rows [N,C] (any strides), dense_init [P,C] contiguous -> [P,C]."""
    return _ScatterRows.apply(rows, pix_idx, n_dev, dense_init)


# ------------------------------------------------------------------------------------------ hash grid
class GridSpec:
    """Per-level constants of the multiresolution hash grid (tcnn HashGrid semantics)."""

    def __init__(self, n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16,
                 per_level_scale=1.447269237440378, n_dims=3):
        import math
        assert n_features == 2, "the HIP kernel is specialised for 2 features per level (dreammat.yaml:47)"
        assert n_dims in (2, 3)
        self.n_levels, self.n_features, self.n_dims = n_levels, n_features, n_dims
        scale, res, size, offset = [], [], [], []
        off = 0
        for l in range(n_levels):
            s = np.float32(math.pow(2.0, l * math.log2(per_level_scale)) * base_resolution - 1.0)
            r = int(math.ceil(float(s))) + 1
            n = min((r ** n_dims + 7) // 8 * 8, 1 << log2_hashmap_size)
            scale.append(float(s)); res.append(r); size.append(n); offset.append(off)
            off += n
        self.total_entries = off
        self.n_params = off * n_features
        self.n_output_dims = n_levels * n_features
        self.c_scale = (ctypes.c_float * n_levels)(*scale)
        self.c_res = (ctypes.c_uint32 * n_levels)(*res)
        self.c_size = (ctypes.c_uint32 * n_levels)(*size)
        self.c_offset = (ctypes.c_uint32 * n_levels)(*offset)
        self.levels = [dict(scale=a, res=b, size=c, offset=d) for a, b, c, d in zip(scale, res, size, offset)]


# backward route: >= this many points -> the binned kernels (dm_hashgrid_bwd_binned), below -> one global atomic pair per corner.
# The workspace (16 B per corner update x 1.3 + the per-split slabs: ~4.3 GB at 8 views x 512^2) is allocated once and kept.
HASHGRID_BINNED_MIN_POINTS = 1 << 16
_HG_WS = {}


def _hashgrid_workspace(nbytes, device):
    key = (device.type, device.index)
    ws = _HG_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        _HG_WS[key] = ws = torch.empty(int(nbytes * 1.1) + 256, dtype=torch.uint8, device=device)
    return ws


class _HashGrid(torch.autograd.Function):
    """x [M,3] (any strides) -> enc, returned as an [M,2L] VIEW of a feature-major [2L,M] buffer.
    `grad_sink`: optional pre-allocated gradient buffer of `table` (the flat all-reduce buffer); the
    backward kernel then scatter-adds straight into it instead of materialising a 50 MB temporary."""

    @staticmethod
    def forward(ctx, x, table, spec, radius, grad_sink):
        _need_cuda(x, table)
        assert table.dtype == torch.float32 and table.is_contiguous()
        M = x.shape[0]
        F = spec.n_output_dims
        enc = torch.empty(F, M, device=x.device, dtype=torch.float32)
        assert x.shape[1] == spec.n_dims
        if M > 0:
            rs, cs = _rs_cs(x)
            fwd, name = ((_lib.lib().dm_hashgrid_fwd, "dm_hashgrid_fwd") if spec.n_dims == 3
                         else (_lib.lib().dm_hashgrid2d_fwd, "dm_hashgrid2d_fwd"))
            # M (12 + levels x 2^d corners x 8 B gathered) read, M x 4 F written -- reported as gather-bound
            with _Timed("hashgrid_fwd", float(M) * (4 * spec.n_dims + spec.n_levels * (1 << spec.n_dims) * 8 + 4 * F)):
                check(fwd(x.data_ptr(), rs, cs, None, M, table.data_ptr(), spec.n_levels, spec.c_scale, spec.c_res, spec.c_size,
                          spec.c_offset, float(radius), enc.data_ptr(), 1, M, _stream()), name)
        ctx.save_for_backward(x, table)
        ctx.spec, ctx.radius, ctx.grad_sink = spec, radius, grad_sink
        return enc.t()

    @staticmethod
    def backward(ctx, g):
        x, table = ctx.saved_tensors
        spec = ctx.spec
        M = x.shape[0]
        sink = ctx.grad_sink
        direct = sink is not None and sink.is_contiguous() and sink.numel() == table.numel()
        dtable = sink if direct else torch.zeros_like(table)
        if M > 0:
            rs, cs = _rs_cs(x)
            grs, gcs = _rs_cs(g)
            L = _lib.lib()
            if spec.n_dims == 2:                       # uv-space field: the plain atomic route
                check(L.dm_hashgrid2d_bwd(x.data_ptr(), rs, cs, None, M, g.data_ptr(), grs, gcs, spec.n_levels, spec.c_scale,
                                          spec.c_res, spec.c_size, spec.c_offset, float(ctx.radius), dtable.data_ptr(), _stream()),
                      "dm_hashgrid2d_bwd")
                return None, (None if direct else dtable), None, None, None
            ws_bytes = int(L.dm_hashgrid_bwd_workspace_bytes(M, spec.n_levels, spec.c_res, spec.c_size)) \
                if M >= HASHGRID_BINNED_MIN_POINTS else 0
            with _Timed(f"hashgrid_bwd[{'binned' if ws_bytes else 'atomic'}]", float(M) * (12 + 4 * spec.n_output_dims)):
                if ws_bytes:
                    ws = _hashgrid_workspace(ws_bytes, x.device)
                    check(L.dm_hashgrid_bwd_binned(x.data_ptr(), rs, cs, None, M, g.data_ptr(), grs, gcs, spec.n_levels,
                                                   spec.c_scale, spec.c_res, spec.c_size, spec.c_offset, float(ctx.radius),
                                                   dtable.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
                          "dm_hashgrid_bwd_binned")
                else:
                    check(L.dm_hashgrid_bwd(x.data_ptr(), rs, cs, None, M, g.data_ptr(), grs, gcs, spec.n_levels,
                                            spec.c_scale, spec.c_res, spec.c_size, spec.c_offset, float(ctx.radius),
                                            dtable.data_ptr(), _stream()), "dm_hashgrid_bwd")
        return None, (None if direct else dtable), None, None, None


def hashgrid_encode(x, table, spec, radius=1.0, grad_sink=None):
    if x.dtype != torch.float32:
        x = x.float()
    return _HashGrid.apply(x, table, spec, radius, grad_sink)


# ------------------------------------------------------------------------------------------ feature network
FIELD_MLP_FUSED = True       # tools/field_mlp_probe.py assigns False to time the torch path the kernels replace; no environment switch


def field_mlp_ok(x_fm, w1, w2):
    """x_fm: feature-major activations [n_in, M] (a transposed view of the hash-grid output)"""
    return (FIELD_MLP_FUSED and x_fm.is_cuda and x_fm.dtype == torch.float32 and x_fm.dim() == 2 and x_fm.stride(1) == 1
            and x_fm.shape[0] in (16, 32) and tuple(w1.shape) == (64, x_fm.shape[0]) and w2.shape[1] == 64 and w2.shape[0] <= 8
            and w1.dtype == torch.float32 and w2.dtype == torch.float32)


class _FieldMlp(torch.autograd.Function):
    """VanillaMLP of the feature field (Linear(n_in, 64, bias=False) -> ReLU -> Linear(64, n_out, bias=False)) in two fused
    kernels (csrc/field_mlp.hip): feature-major in and out, the hidden layer recomputed in the backward."""

    @staticmethod
    def forward(ctx, x_fm, w1, w2):
        n_in, M = x_fm.shape
        n_out = w2.shape[0]
        w1c, w2c = w1.contiguous(), w2.contiguous()
        Mp = (M + 3) // 4 * 4                  # channel pitch: the shade kernels stream a feature channel 16 bytes per lane
        y = torch.empty(n_out, Mp, device=x_fm.device, dtype=torch.float32)[:, :M]
        if M > 0:
            with _Timed("field_mlp_fwd", 2.0 * M * (n_in * 64 + 64 * n_out)):
                check(_lib.lib().dm_field_mlp_fwd(x_fm.data_ptr(), x_fm.stride(0), M, w1c.data_ptr(), w2c.data_ptr(), n_in, n_out,
                                                  y.data_ptr(), Mp, _stream()), "dm_field_mlp_fwd")
        ctx.save_for_backward(x_fm, w1c, w2c)
        return y

    @staticmethod
    def backward(ctx, g):
        x_fm, w1, w2 = ctx.saved_tensors
        n_in, M = x_fm.shape
        n_out = w2.shape[0]
        dx = torch.empty(n_in, M, device=g.device, dtype=torch.float32)
        dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
        if M > 0:
            with _Timed("field_mlp_bwd", 2.0 * M * (3 * n_in * 64 + 3 * 64 * n_out)):
                # g [n_out, M] by strides: element (k, m) at g.stride(1) * m + g.stride(0) * k
                check(_lib.lib().dm_field_mlp_bwd(x_fm.data_ptr(), x_fm.stride(0), M, w1.data_ptr(), w2.data_ptr(), n_in, n_out,
                                                  g.data_ptr(), g.stride(1), g.stride(0), dx.data_ptr(), M, dw1.data_ptr(),
                                                  dw2.data_ptr(), _stream()), "dm_field_mlp_bwd")
        return dx, dw1, dw2


def field_mlp(x_fm, w1, w2):
    """[n_in, M] -> [n_out, M] (both feature-major, fp32)."""
    _need_cuda(x_fm, w1, w2)
    return _FieldMlp.apply(x_fm, w1, w2)


# ------------------------------------------------------------------------------------------ shading
SHADE_DUMP = {"path": None}      # measurement aid (bench.py --dump-shade): the next shade forward saves its inputs here, once
SHADE_KEEP = {"on": False, "last": None}     # measurement aid (bench.py): keep the arguments of the last shade forward for a replay


class _Shade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, nrm, view, pix_idx, n_dev, env_of_view, atlas, mat, HW, want_debug):
        _need_cuda(feat, nrm, view, pix_idx, env_of_view)
        N = feat.shape[0]
        dev = feat.device
        if SHADE_DUMP["path"]:
            torch.save({"feat": feat.detach().cpu(), "nrm": nrm.cpu(), "view": view.cpu(), "pix_idx": pix_idx.cpu(),
                        "env_of_view": env_of_view.cpu(), "HW": HW, "mat": [mat.min_metallic, mat.max_metallic, mat.min_roughness,
                                                                             mat.max_roughness],
                        "spec_packed": atlas.spec_packed.cpu(), "diff_packed": atlas.diff_packed.cpu(), "fg_lut": atlas.fg_lut.cpu(),
                        "atlas_fields": {k: (list(getattr(atlas.struct, k)) if k in ("mip_off", "mip_res") else getattr(atlas.struct, k))
                                         for k in ("spec_env_stride", "diff_env_stride", "mip_off", "mip_res", "n_mips", "diff_res",
                                                   "lut_res", "min_rough_mip", "max_rough_mip", "texel_format")}}, SHADE_DUMP["path"])
            SHADE_DUMP["path"] = None
        if SHADE_KEEP["on"]:
            SHADE_KEEP["last"] = (feat.detach(), nrm, view, pix_idx, n_dev, env_of_view, atlas, mat, HW)
        Np = (N + 3) // 4 * 4                   # channel pitch: the kernel stores 16 bytes (4 rows) per lane
        color = torch.empty(3, Np, device=dev)[:, :N]
        dbg = [None] * 7
        if want_debug:
            dbg = [torch.empty(N, 3, device=dev) for _ in range(5)] + [torch.empty(N, 1, device=dev) for _ in range(2)]
        if N > 0:
            with _Timed("shade_fwd" + ("+dbg" if want_debug else ""), 56.0 * N):
                check(_lib.lib().dm_shade_fwd(ctypes.byref(atlas.struct), ctypes.byref(mat), nrm.data_ptr(),
                                              *_rs_cs(nrm), view.data_ptr(), *_rs_cs(view), feat.data_ptr(),
                                              *_rs_cs(feat), pix_idx.data_ptr(), env_of_view.data_ptr(),
                                              n_dev.data_ptr(), N, HW, env_of_view.numel(), color.data_ptr(), 1, Np,
                                              *[d.data_ptr() if d is not None else None for d in dbg], _stream()),
                      "dm_shade_fwd")
        ctx.save_for_backward(feat, nrm, view, pix_idx, n_dev, env_of_view)
        ctx.atlas, ctx.mat, ctx.HW = atlas, mat, HW
        outs = (color.t(),) + tuple(d for d in dbg if d is not None)
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g, *unused):
        feat, nrm, view, pix_idx, n_dev, env_of_view = ctx.saved_tensors
        N = feat.shape[0]
        Np = (N + 3) // 4 * 4
        dfeat = torch.zeros(5, Np, device=feat.device)[:, :N]
        if N > 0:
            with _Timed("shade_bwd", 76.0 * N):
                check(_lib.lib().dm_shade_bwd(ctypes.byref(ctx.atlas.struct), ctypes.byref(ctx.mat), nrm.data_ptr(),
                                              *_rs_cs(nrm), view.data_ptr(), *_rs_cs(view), feat.data_ptr(),
                                              *_rs_cs(feat), pix_idx.data_ptr(), env_of_view.data_ptr(),
                                              n_dev.data_ptr(), N, ctx.HW, env_of_view.numel(), g.data_ptr(), *_rs_cs(g),
                                              dfeat.data_ptr(),
                                              1, Np, _stream()), "dm_shade_bwd")
        return (dfeat.t(),) + (None,) * 9


def shade(feat, nrm, view, pix_idx, n_dev, env_of_view, atlas, mat, HW, want_debug=True):
    """-> (color [N,3], albedo, spec_light, diff_light, spec_color, diff_color, metallic, roughness)."""
    return _Shade.apply(feat, nrm, view, pix_idx, n_dev, env_of_view, atlas, mat, HW, want_debug)


class _MatReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, featj, n_dev):
        _need_cuda(feat, featj)
        N = feat.shape[0]
        loss = torch.zeros(1, device=feat.device)
        if N > 0:
            check(_lib.lib().dm_matreg_fwd(feat.data_ptr(), *_rs_cs(feat), featj.data_ptr(), *_rs_cs(featj),
                                           n_dev.data_ptr(), N, loss.data_ptr(), _stream()), "dm_matreg_fwd")
        ctx.save_for_backward(feat, featj, n_dev)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        feat, featj, n_dev = ctx.saved_tensors
        N = feat.shape[0]
        df = torch.zeros(5, N, device=feat.device)
        dj = torch.zeros(5, N, device=feat.device)
        if N > 0:
            check(_lib.lib().dm_matreg_bwd(feat.data_ptr(), *_rs_cs(feat), featj.data_ptr(), *_rs_cs(featj),
                                           n_dev.data_ptr(), N, 1.0, df.data_ptr(), 1, N, dj.data_ptr(),
                                           1, N, _stream()), "dm_matreg_bwd")
        return (df * g).t(), (dj * g).t(), None   # upstream scalar applied on device: no host sync


def material_smoothness(feat, featj, n_dev):
    return _MatReg.apply(feat, featj, n_dev)


# ------------------------------------------------------------------------------------------ attention
def attention_select(name=None):
    """Kernel family of every later attention() call: "auto" (default) | "w128" (one wave per SIMD, 128 query rows per wave; 64-wide heads, Sq % 512 == 0) | "w64" (one wave per SIMD; 64-wide heads, whole kv
    tiles) | "v3l" | "staged"; None restores the DREAMMAT_ATTN_KERNEL / default choice (dm_attention_select)."""
    check(_lib.lib().dm_attention_select(name.encode() if name is not None else None), "dm_attention_select")


def attention_variant():
    """name of the family the library dispatches from (dm_attention_selected: what dm_attention_select / the environment chose)."""
    return _lib.lib().dm_attention_selected().decode()


def attention(q, k, vt, heads, scale=None):
    """q [B,Sq,C], k [B,Skv,C] bf16 (C = heads*D, last dim contiguous), vt [B,C,Skv_pad] bf16
    (V transposed, rows zero-padded to a multiple of 8) -> out [B,Sq,C] bf16."""
    _need_cuda(q, k, vt)
    fn, name = _sym("dm_attention_fwd_bf16", _same_half(q, k, vt))
    B, Sq, C = q.shape
    Skv = k.shape[1]
    D = C // heads
    assert q.stride(2) == 1 and k.stride(2) == 1 and vt.stride(2) == 1
    out = torch.empty(B, Sq, C, device=q.device, dtype=q.dtype)
    sc = float(scale) if scale is not None else float(D) ** -0.5
    with _Timed(f"attention_fwd_bf16[Sq={Sq},Skv={Skv},h={heads},D={D}]", 4.0 * B * Sq * Skv * C):
        check(fn(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), B, heads,
                                               Sq, Skv, D, q.stride(0), q.stride(1), D, k.stride(0), k.stride(1), D,
                                               vt.stride(0), D * vt.stride(1), vt.stride(1), out.stride(0),
                 out.stride(1), D, sc, _stream()), name)
    return out


_FP8_WS = {}
_FP8_WS_RETIRED = []


def attention_fp8_ok(q, k, heads):
    """dm_attention_fwd_fp8 serves this call: 64-wide heads, whole 256-row query blocks and 64-row kv tiles"""
    D = q.shape[-1] // heads
    return q.is_cuda and q.dtype in HALF_DTYPES and D == 64 and q.shape[1] % 256 == 0 and k.shape[1] % 64 == 0


def attention_fp8(q, k, vt, heads, scale=None):
    """attention() with both matrix products on the MX-FP8 matrix instruction (csrc/attn_fp8.hip: BASELINE configs[4]); the same
    16-bit tensors in and out, the 8-bit operands live in a per-device scratch buffer (stream order, grow-only).
    SINGLE-STREAM ASSUMPTION: the scratch buffer is shared by every call on the device and baked into captured
    graphs; two calls are ordered only by the stream they run on.  The nets of this repo run on one stream (the capture of
    guidance.hip_graph replays on that same stream; its warm-up side stream is joined before the capture starts), so nothing
    overlaps; a caller that runs fp8 attention on two streams at once must serialise them with events itself."""
    _need_cuda(q, k, vt)
    dt = _same_half(q, k, vt)
    assert attention_fp8_ok(q, k, heads) and q.stride(2) == 1 and k.stride(2) == 1 and vt.stride(2) == 1
    B, Sq, C = q.shape
    Skv = k.shape[1]
    D = C // heads
    need = int(_lib.lib().dm_attention_fp8_workspace_bytes(B, heads, Sq, Skv))
    ws = _FP8_WS.get(q.device)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.DmError("attention_fp8: the 8-bit operand workspace must exist before a stream capture (run the shape eagerly once)")
        if ws is not None:
            _FP8_WS_RETIRED.append(ws)         # never freed: a captured hipGraph (guidance `hip_graph`) replays launches holding its address
        ws = _FP8_WS[q.device] = torch.empty(max(need, 2 * (ws.numel() if ws is not None else 0)), device=q.device, dtype=torch.uint8)
    out = torch.empty(B, Sq, C, device=q.device, dtype=dt)
    sc = float(scale) if scale is not None else float(D) ** -0.5
    with _Timed(f"attention_fwd_fp8[Sq={Sq},Skv={Skv},h={heads},D={D}]", 4.0 * B * Sq * Skv * C):
        check(_lib.lib().dm_attention_fwd_fp8(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), B, heads, Sq, Skv, D,
                                              q.stride(0), q.stride(1), D, k.stride(0), k.stride(1), D, vt.stride(0),
                                              D * vt.stride(1), vt.stride(1), out.stride(0), out.stride(1), D, sc,
                                              1 if dt == torch.float16 else 0, ws.data_ptr(), ws.numel(), _stream()),
              "dm_attention_fwd_fp8")
    return out


ATTN_BWD_MAX_D = 128     # dm_attention_bwd_bf16 (csrc/attn_bwd.hip): head sizes up to 128


def attention_train_ok(q, k, v, heads):
    """the differentiated MFMA attention serves this call (bf16 on the device, head size a multiple of 8 up to 128,
    16 B aligned rows); otherwise the caller composes the product from matmuls under autograd."""
    D = q.shape[-1] // heads
    return (q.is_cuda and q.dtype == k.dtype == v.dtype == torch.bfloat16 and D % 8 == 0 and D <= ATTN_BWD_MAX_D
            and k.shape == v.shape and q.shape[0] * heads <= 65535)


def _attention_fwd_lse(q, k, v, heads, scale):
    """-> (q, k, v as handed to the kernel, out [B,Sq,C] bf16, lse [B,heads,Sq] fp32 = rowmax + log2(rowsum), log2 domain)."""
    B, Sq, C = q.shape
    Skv, D = k.shape[1], C // heads
    q = q if q.stride(2) == 1 and q.stride(1) % 8 == 0 and q.stride(0) % 8 == 0 else q.contiguous()
    if not (k.stride(2) == 1 and k.stride() == v.stride() and k.stride(1) % 8 == 0 and k.stride(0) % 8 == 0):
        k, v = k.contiguous(), v.contiguous()
    out = torch.empty(B, Sq, C, device=q.device, dtype=torch.bfloat16)
    lse = torch.empty(B, heads, Sq, device=q.device, dtype=torch.float32)
    with _Timed(f"attention_fwd_lse_bf16[Sq={Sq},Skv={Skv},h={heads},D={D}]", 4.0 * B * Sq * Skv * C):
        check(_lib.lib().dm_attention_fwd_lse_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                                   B, heads, Sq, Skv, D, q.stride(0), q.stride(1), D, k.stride(0),
                                                   k.stride(1), D, out.stride(0), out.stride(1), D, scale, _stream()),
              "dm_attention_fwd_lse_bf16")
    return q, k, v, out, lse


def attention_fwd_lse(q, k, v, heads, scale):
    """forward half of attention_train on its own (tests): -> (out, lse)."""
    _need_cuda(q, k, v)
    return _attention_fwd_lse(q, k, v, heads, float(scale))[3:]


class _AttentionTrain(torch.autograd.Function):
    """softmax(q k^T / sqrt(D)) v under autograd: forward = the generic MFMA kernel, which also writes one statistic per
    query row; backward = dm_attention_bwd_bf16 (probabilities recomputed tile by tile, nothing S x S stored)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        q, k, v, out, lse = _attention_fwd_lse(q, k, v, heads, scale)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.heads, ctx.scale = heads, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        heads = ctx.heads
        B, Sq, C = q.shape
        Skv, D = k.shape[1], C // heads
        # q, out, dout and dq share one set of strides in the kernel (k, v, dk, dv another)
        qc = q if q.is_contiguous() else q.contiguous()
        kc, vc = (k, v) if k.is_contiguous() and v.is_contiguous() else (k.contiguous(), v.contiguous())
        dout = dout.to(torch.bfloat16).contiguous()
        dq, dk, dv = torch.empty_like(qc), torch.empty_like(kc), torch.empty_like(vc)
        delta = torch.empty_like(lse)
        with _Timed(f"attention_bwd_bf16[Sq={Sq},Skv={Skv},h={heads},D={D}]", 14.0 * B * Sq * Skv * C):
            check(_lib.lib().dm_attention_bwd_bf16(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                                   lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                                   dv.data_ptr(), B, heads, Sq, Skv, D, Sq * C, C, D, Skv * C, C, D,
                                                   ctx.scale, _stream()), "dm_attention_bwd_bf16")
        return dq, dk, dv, None, None


def attention_train(q, k, v, heads, scale=None):
    """differentiable attention: q [B,Sq,C], k, v [B,Skv,C] bf16 (C = heads*D) -> [B,Sq,C] bf16."""
    _need_cuda(q, k, v)
    assert attention_train_ok(q, k, v, heads), "attention_train: bf16 device tensors with head size % 8 == 0, <= 128"
    sc = float(scale) if scale is not None else float(q.shape[-1] // heads) ** -0.5
    return _AttentionTrain.apply(q, k, v, heads, sc)


# ------------------------------------------------------------------------------------------ convolution
CONV_MAX_TENSOR_BYTES = 0xffffff00      # per-launch limit of the buffer-addressed kernels (tests lower it)


def conv3x3_nhwc(x_nhwc, w_tap_major, bias, stride=1, pad=(1, 1), out_hw=None, rowbias=None, residual=None):
    """x [B,H,W,Cin] bf16 contiguous, w [Cout, 9*Cin] bf16 (tap-major) -> y [B,Ho,Wo,Cout] bf16.
    rowbias [B,Cout] / residual [B,Ho,Wo,Cout] (bf16) are added in the kernel epilogue (Cin % 64 == 0)."""
    _need_cuda(x_nhwc, w_tap_major, rowbias, residual)
    assert x_nhwc.is_contiguous() and w_tap_major.is_contiguous()
    fn, name = _sym("dm_conv3x3_nhwc_bf16_fused", _same_half(x_nhwc, w_tap_major, bias, rowbias, residual))
    B, H, W, Cin = x_nhwc.shape
    Cout = w_tap_major.shape[0]
    if out_hw is None:
        out_hw = ((H + 2 * pad[0] - 3) // stride + 1, (W + 2 * pad[1] - 3) // stride + 1)
    Ho, Wo = out_hw
    y = torch.empty(B, Ho, Wo, Cout, device=x_nhwc.device, dtype=x_nhwc.dtype)
    if rowbias is not None:
        assert rowbias.is_contiguous() and tuple(rowbias.shape) == (B, Cout)
    if residual is not None:
        assert residual.is_contiguous() and residual.shape == y.shape
    # the LDS-DMA kernel addresses a tensor with 32-bit byte offsets: batches whose activations pass 4 GB (16 views at
    # 1024^2 through the VAE) run as consecutive image chunks of the same contiguous buffers
    per_img = 2 * max(H * W * Cin, Ho * Wo * Cout)
    if B > 1 and B * per_img > CONV_MAX_TENSOR_BYTES:
        step = max(1, CONV_MAX_TENSOR_BYTES // per_img)
        for b0 in range(0, B, step):
            b1 = min(B, b0 + step)
            y[b0:b1] = conv3x3_nhwc(x_nhwc[b0:b1], w_tap_major, bias, stride, pad, out_hw,
                                    rowbias[b0:b1] if rowbias is not None else None,
                                    residual[b0:b1] if residual is not None else None)
        return y
    with _Timed(f"conv3x3[{Cin}->{Cout}@{Ho}x{Wo},s{stride}]", 2.0 * B * Ho * Wo * Cout * 9 * Cin):
        check(fn(
            x_nhwc.data_ptr(), w_tap_major.data_ptr(), bias.data_ptr() if bias is not None else None,
            rowbias.data_ptr() if rowbias is not None else None, residual.data_ptr() if residual is not None else None,
            y.data_ptr(), B, H, W, Cin, Ho, Wo, Cout, stride, pad[0], pad[1], _stream()), name)
    return y


SMALL_CONV_CIN = (4, 8, 16, 22, 32)


def conv3x3_small_nhwc(x_nhwc, w_tap_major, bias, stride=1, pad=(1, 1), act=0, residual=None):
    """few-channel stem convs with bias, optional residual and SiLU fused (forward only; dm_conv3x3_small_res_nhwc_bf16): the patch
    kernel on the matrix pipe for Cin even and <= 32 (or Cin = 128 with Cout <= 32: a data gradient) and Cout % 4 == 0 at stride
    1 | 2, the direct kernel behind it for Cin in SMALL_CONV_CIN, Cout % 16 == 0.
    residual [Br, Ho, Wo, Cout] (B % Br == 0): added before the rounding, image b takes residual image b % Br."""
    _need_cuda(x_nhwc, w_tap_major, bias, residual)
    assert x_nhwc.is_contiguous() and w_tap_major.is_contiguous()
    fn, name = _sym("dm_conv3x3_small_res_nhwc_bf16", _same_half(x_nhwc, w_tap_major, bias, residual))
    B, H, W, Cin = x_nhwc.shape
    Cout = w_tap_major.shape[0]
    Ho, Wo = (H + 2 * pad[0] - 3) // stride + 1, (W + 2 * pad[1] - 3) // stride + 1
    Br = 0
    if residual is not None:
        Br = residual.shape[0]
        assert residual.is_contiguous() and tuple(residual.shape[1:]) == (Ho, Wo, Cout) and B % Br == 0
    y = torch.empty(B, Ho, Wo, Cout, device=x_nhwc.device, dtype=x_nhwc.dtype)
    with _Timed(f"conv3x3_small[{Cin}->{Cout}@{Ho}x{Wo},s{stride}]", 2.0 * B * Ho * Wo * (Cin + Cout * (2 if Br else 1))):
        check(fn(x_nhwc.data_ptr(), w_tap_major.data_ptr(), bias.data_ptr() if bias is not None else None,
                 residual.data_ptr() if Br else None, Br, y.data_ptr(), B, H, W, Cin, Ho, Wo, Cout, stride, pad[0], pad[1],
                 int(act), _stream()), name)
    return y


def gemm_fused(x, w, bias=None, residual=None, geglu=False, out=None):
    """y[..., N] = x[..., K] @ w[N, K]^T + bias (+ residual[..., N]) on the 1-tap LDS-DMA kernel (forward only).
    geglu: `w` / `bias` rows interleaved by `geglu_interleave`; returns value * gelu(gate), [..., N/2].
    out: a contiguous tensor of the result's shape and dtype to write into (a reused workspace)."""
    _need_cuda(x, w, bias, residual)
    assert x.is_contiguous() and w.is_contiguous()
    fn, name = _sym("dm_gemm_bf16_fused", _same_half(x, w, bias, residual))
    N, K = w.shape
    M = x.numel() // K
    assert x.shape[-1] == K
    No = N // 2 if geglu else N
    if out is not None:
        assert out.is_contiguous() and out.dtype == x.dtype and out.numel() == M * No and 2 * M * max(K, No) <= CONV_MAX_TENSOR_BYTES
        y = out
    else:
        y = torch.empty(*x.shape[:-1], No, device=x.device, dtype=x.dtype)
    if 2 * M * max(K, No) > CONV_MAX_TENSOR_BYTES:           # same 32-bit addressing limit: row chunks of the same buffers
        rows = max(16, (CONV_MAX_TENSOR_BYTES // (2 * max(K, No))) // 16 * 16)
        x2, y2 = x.reshape(M, K), y.view(M, No)
        r2 = residual.reshape(M, No) if residual is not None else None
        for m0 in range(0, M, rows):
            m1 = min(M, m0 + rows)
            y2[m0:m1] = gemm_fused(x2[m0:m1], w, bias, r2[m0:m1] if r2 is not None else None, geglu)
        return y
    if bias is not None:
        assert bias.is_contiguous() and bias.numel() == N
    if residual is not None:
        assert residual.is_contiguous() and residual.numel() == y.numel()
    with _Timed(f"gemm{'+geglu' if geglu else ''}{'+res' if residual is not None else ''}[M={M},K={K},N={N}]", 2.0 * M * K * N):
        check(fn(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                 residual.data_ptr() if residual is not None else None, y.data_ptr(), M, K, N, 1 if geglu else 0, _stream()), name)
    return y


def gemm_batched(x, w, out):
    """out[i] = x[i] @ w[i]^T for every item of the leading dimension (dm_gemm_*_batched): x [G, M, K], w [G, N, K], out [G, M, N]
    contiguous 16-bit, M % 256 == 0, K % 64 == N % 64 == 0 (forward only)."""
    _need_cuda(x, w, out)
    G, M, K = x.shape
    N = w.shape[1]
    assert x.is_contiguous() and w.is_contiguous() and out.is_contiguous() and w.shape == (G, N, K) and out.numel() == G * M * N
    fn, name = _sym("dm_gemm_bf16_batched", _same_half(x, w, out))
    with _Timed(f"gemm_batched[G={G},M={M},K={K},N={N}]", 2.0 * G * M * K * N):
        check(fn(x.data_ptr(), w.data_ptr(), out.data_ptr(), G, M, K, N, _stream()), name)
    return out


def linear_small_ok(M, K, N):
    return K == 8 and N == 8        # (8 -> 8: the data gradient is the same shape on w^T)


class _LinearSmallFrozen(torch.autograd.Function):
    """few-channel frozen Linear (AutoencoderKL's quant_conv, 8 -> 8) under autograd: dm_linear_small forward, the same on w^T back"""

    @staticmethod
    def forward(ctx, x, w, w_t, bias):
        ctx.w_t = w_t
        return linear_small(x, w, bias)

    @staticmethod
    def backward(ctx, g):
        return (linear_small(g.contiguous(), ctx.w_t, None) if ctx.needs_input_grad[0] else None), None, None, None


def linear_small(x, w, bias):
    """x [..., 8] contiguous, w [8, 8] -> [..., 8] (forward only; linear_small_autograd wraps it)"""
    _need_cuda(x, w, bias)
    N, K = w.shape
    M = x.numel() // K
    assert x.is_contiguous() and w.is_contiguous() and linear_small_ok(M, K, N)
    y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=x.dtype)
    fn, name = _sym("dm_linear_small_bf16", _same_half(x, w, bias))
    with _Timed(f"linear_small[M={M},K={K},N={N}]", 2.0 * M * K * N):
        check(fn(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), M, K, N, _stream()), name)
    return y


def linear_small_autograd(x, w, w_t, bias):
    return _LinearSmallFrozen.apply(x, w, w_t, bias)


class _LinearFrozen(torch.autograd.Function):
    """y = x @ w^T + bias (+ residual) with FROZEN w / bias, differentiable wrt x (and the residual): forward and data gradient on
    the fused GEMM kernel (round 6: the 1 x 1 shortcut convolutions and the attention projections of the differentiated VAE
    encoder, dreammat_guidance.py:284-292, ran on ATen / hipBLASLt under autograd).  dx = g @ w: the same kernel on w^T."""

    @staticmethod
    def forward(ctx, x, w, w_t, bias, residual):
        ctx.w_t = w_t
        return gemm_fused(x, w, bias, residual)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dx = gemm_fused(g, ctx.w_t, None, None) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, (g if ctx.needs_input_grad[4] else None)


def linear_frozen_autograd(x, w, w_t, bias=None, residual=None):
    """x [..., K] contiguous, w [N, K], w_t = w.t().contiguous() [K, N] (prepared once by the caller); see _LinearFrozen."""
    return _LinearFrozen.apply(x, w, w_t, bias, residual)


def gemm_fused_ok(M, K, N, geglu=False):
    return M % 16 == 0 and K % 64 == 0 and N % (128 if geglu else 64) == 0


def geglu_interleave(t):
    """rows [value(inner) | gate(inner)] -> blocks of 32 value rows followed by their 32 gate rows (dm_gemm_bf16_fused)."""
    inner = t.shape[0] // 2
    assert inner % 32 == 0
    v = t[:inner].reshape(inner // 32, 32, *t.shape[1:])
    g = t[inner:].reshape(inner // 32, 32, *t.shape[1:])
    return torch.stack([v, g], dim=1).reshape(t.shape).contiguous()


class _Conv3x3S1(torch.autograd.Function):
    """stride-1, pad-1 3x3 conv with frozen weights: backward = the same kernel on the flipped weights."""

    @staticmethod
    def forward(ctx, x_nhwc, w_fwd, w_dgrad, bias, residual):
        ctx.w_dgrad = w_dgrad
        return conv3x3_nhwc(x_nhwc, w_fwd, bias, 1, (1, 1), None, None, residual)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dx = conv3x3_nhwc(g, ctx.w_dgrad, None, 1, (1, 1)) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, (g if ctx.needs_input_grad[4] else None)


def conv3x3_s1_autograd(x_nhwc, w_fwd, w_dgrad, bias, residual=None):
    """y = conv(x) + bias (+ residual, folded into the kernel epilogue); differentiable wrt x and residual."""
    return _Conv3x3S1.apply(x_nhwc, w_fwd, w_dgrad, bias, residual)


class _ConvStemS1(torch.autograd.Function):
    """stride-1, pad-1 3x3 conv of a FEW-channel image (Cin <= 4) with frozen weights, differentiable wrt the image: the VAE
    encoder's conv_in under SDS (dreammat_guidance.py:284-292 -- the render is the leaf).  Forward = the stem kernel on the image
    padded to 4 channels (the im2col + GEMM + bias-add lowering it replaces wrote a 9x copy of the image and took three passes
    over the 128-channel output); backward = the same kernel on the flipped weights, Cout -> 4 channels (the patch form of
    dm_conv3x3_small_nhwc_bf16 takes 128 input channels), instead of a [B HW, 128] x [128, 27] product + col2im."""

    @staticmethod
    def forward(ctx, x_nhwc, w4, w_dgrad4, bias):
        B, H, W, Cin = x_nhwc.shape
        x4 = torch.nn.functional.pad(x_nhwc, (0, 4 - Cin)) if Cin < 4 else x_nhwc.contiguous()
        ctx.w_dgrad4, ctx.cin = w_dgrad4, Cin
        return conv3x3_small_nhwc(x4, w4, bias, 1, (1, 1), 0)

    @staticmethod
    def backward(ctx, g):
        dx4 = conv3x3_small_nhwc(g.contiguous(), ctx.w_dgrad4, None, 1, (1, 1), 0)        # [B, H, W, 4]
        return dx4[..., :ctx.cin], None, None, None


def conv3x3_stem_autograd(x_nhwc, w4, w_dgrad4, bias):
    return _ConvStemS1.apply(x_nhwc, w4, w_dgrad4, bias)


# the 2 x 2 kernel stores its four channel blocks as the sub-pixels of the finer tensor (round 5); tests assign False for the
# interleaving-copy form it replaces
SUBPIXEL_STORE = True


def conv2x2_nhwc(x_nhwc, w4, bias, out_hw, pad=(1, 1), label="conv2x2", subpixel=0):
    """dm_conv2x2_nhwc_bf16: y[b, yo, xo, n] = sum over the 2 x 2 window from (yo - pad, xo - pad) of x . w4[n][2 dy + dx]; x [B,H,W,Cin],
    w4 [Cout, 4 * Cin], y [B, Ho, Wo, Cout] bf16.  Like conv3x3_nhwc, batches whose tensors pass the kernel's 32-bit byte offsets
    run as consecutive image chunks.
    subpixel (dm_conv2x2_subpixel_nhwc_bf16): the four Cout / 4-channel blocks of a pixel are stored as the four sub-pixels of the 2x
    finer tensor -- 1: [B, 2 Ho, 2 Wo, Cout / 4], block 2 py + px of (u, v) at (2u + py, 2v + px); 2: [B, 2 (Ho - 1), 2 (Wo - 1),
    Cout / 4], block 2 py + px of grid position (u, v) at (2u - py, 2v - px) where that is inside."""
    _need_cuda(x_nhwc, w4, bias)
    assert x_nhwc.is_contiguous() and w4.is_contiguous() and subpixel in (0, 1, 2)
    fn, name = _sym("dm_conv2x2_subpixel_nhwc_bf16" if subpixel else "dm_conv2x2_nhwc_bf16", _same_half(x_nhwc, w4, bias))
    B, H, W, Cin = x_nhwc.shape
    Cout = w4.shape[0]
    Ho, Wo = out_hw
    if subpixel:
        assert (Cout // 4) % 16 == 0
        Hd, Wd = (2 * Ho, 2 * Wo) if subpixel == 1 else (2 * (Ho - 1), 2 * (Wo - 1))
        y = torch.empty(B, Hd, Wd, Cout // 4, device=x_nhwc.device, dtype=x_nhwc.dtype)
    else:
        y = torch.empty(B, Ho, Wo, Cout, device=x_nhwc.device, dtype=x_nhwc.dtype)
    per_img = 2 * max(H * W * Cin, Ho * Wo * Cout)
    step = B if B * per_img <= CONV_MAX_TENSOR_BYTES else max(1, CONV_MAX_TENSOR_BYTES // per_img)
    for b0 in range(0, B, step):
        b1 = min(B, b0 + step)
        with _Timed(f"{label}[{Cin}->{Cout}@{Ho}x{Wo}]", 2.0 * (b1 - b0) * Ho * Wo * 4.0 * Cin * Cout):
            args = (x_nhwc[b0:b1].data_ptr(), w4.data_ptr(), bias.data_ptr() if bias is not None else None,
                    y[b0:b1].data_ptr(), b1 - b0, H, W, Cin, Ho, Wo, Cout, pad[0], pad[1])
            check(fn(*args, subpixel, _stream()) if subpixel else fn(*args, _stream()), name)
    return y


def subpixel_upsample_weights(w, bias):
    """nearest-2x upsampling followed by a 3x3 convolution (pad 1) with weights w [Cout, Cin, 3, 3], as ONE 2 x 2 convolution at
    the source resolution: output (2u + py, 2v + px) reads source rows (2u + py + ky - 1) >> 1 = u + py - 1 + dy with dy in {0, 1}:
    taps ky in S(py, dy) fall on window row dy, S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}; likewise in x.
    -> (w4 [4 * Cout, 4 * Cin] for dm_conv2x2_nhwc_bf16: output channel (py, px, co), tap (dy, dx); bias tiled x 4)."""
    Cout, Cin = w.shape[:2]
    S = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
    wf = w.float()
    w4 = wf.new_zeros(2, 2, Cout, 2, 2, Cin)                    # [py, px, co, dy, dx, ci]
    for (py, dy), kys in S.items():
        for (px, dx), kxs in S.items():
            acc = 0
            for ky in kys:
                for kx in kxs:
                    acc = acc + wf[:, :, ky, kx]
            w4[py, px, :, dy, dx, :] = acc
    b4 = bias.repeat(4).contiguous() if bias is not None else None
    return w4.reshape(4 * Cout, 4 * Cin).to(w.dtype).contiguous(), b4


def conv3x3_upsampled_nhwc(x_nhwc, w4, b4):
    """conv3x3(pad 1)(nearest-2x upsample(x)) from subpixel_upsample_weights: x [B,h,w,Cin] -> [B,2h,2w,Cout] (forward only)."""
    _need_cuda(x_nhwc, w4, b4)
    assert x_nhwc.dtype in HALF_DTYPES and x_nhwc.is_contiguous() and w4.is_contiguous()
    B, h, w, Cin = x_nhwc.shape
    C4 = w4.shape[0]
    Cout = C4 // 4
    if SUBPIXEL_STORE and Cout % 16 == 0 and h >= 1 and w >= 1:
        # the kernel stores parity (py, px) of grid position (u + py, v + px) at output pixel (2u + py, 2v + px) itself
        return conv2x2_nhwc(x_nhwc, w4, b4, (h + 1, w + 1), (1, 1), "conv2x2_upsample", subpixel=2)
    y = conv2x2_nhwc(x_nhwc, w4, b4, (h + 1, w + 1), (1, 1), "conv2x2_upsample")
    out = torch.empty(B, h, 2, w, 2, Cout, device=x_nhwc.device, dtype=x_nhwc.dtype)
    for py in range(2):
        for px in range(2):
            blk = (2 * py + px) * Cout
            out[:, :, py, :, px] = y[:, py:py + h, px:px + w, blk:blk + Cout]
    return out.view(B, 2 * h, 2 * w, Cout)


def subpixel_dgrad_weights(w_fwd, Cin):
    """weights of dm_conv2x2_nhwc_bf16 for the data gradient of `conv3x3(F.pad(x, (0,1,0,1)), stride=2)` with forward weights
    w_fwd [Cout, 9 * Cin] (tap-major): [4 * Cin, 4 * Cout], output channel (py, px, ci), tap (dy, dx) of the 2 x 2 window over
    g[u - 1 .. u, v - 1 .. v].  dx[2u + py, ..] takes g[u] through ky = py (dy = 1) and, for py = 0 only, g[u - 1] through ky = 2
    (dy = 0); likewise in x."""
    Cout = w_fwd.shape[0]
    w = w_fwd.view(Cout, 3, 3, Cin)
    ws = w.new_zeros(2, 2, Cin, 2, 2, Cout)                     # [py, px, ci, dy, dx, co]
    ky = {(0, 1): 0, (0, 0): 2, (1, 1): 1}                      # (parity, window row) -> forward tap row
    for (py, dy), kyy in ky.items():
        for (px, dx), kxx in ky.items():
            ws[py, px, :, dy, dx, :] = w[:, kyy, kxx, :].t()
    return ws.reshape(4 * Cin, 4 * Cout).contiguous()


class _Conv3x3S2(torch.autograd.Function):
    """stride-2 3x3 conv with leading pad p (0: AutoencoderKL's F.pad(0,1,0,1) downsampler; 1: UNet's) and frozen
    weights.  Backward, p = 0 with even sizes and whole 64-channel blocks: the sub-pixel form -- ONE 2 x 2 convolution at the
    gradient's resolution whose 4 Cin output channels are the four parities of dx (dm_conv2x2_nhwc_bf16, 16 tap-blocks per
    gradient pixel), then the interleave.  Otherwise: the stride-1 3x3 kernel on the zero-inserted gradient (G_up[2i,2j] = g[i,j])
    with leading pad 2-p and the flipped / channel-swapped weights (36 tap-blocks and a zero tensor 4x the gradient's size)."""

    @staticmethod
    def forward(ctx, x_nhwc, w_fwd, w_dgrad, bias, p, w_sub=None):
        B, H, W, Cin = x_nhwc.shape
        Ho, Wo = (H + 2 * p - 3 + (1 - p)) // 2 + 1, (W + 2 * p - 3 + (1 - p)) // 2 + 1   # p=0: trailing pad 1
        ctx.w_dgrad, ctx.p, ctx.hw = w_dgrad, p, (H, W)
        ctx.w_sub, ctx.cin = w_sub, Cin
        return conv3x3_nhwc(x_nhwc, w_fwd, bias, 2, (p, p), (Ho, Wo))

    @staticmethod
    def backward(ctx, g):
        B, Ho, Wo, C = g.shape
        H, W = ctx.hw
        Cin = ctx.cin
        ws = ctx.w_sub                          # subpixel_dgrad_weights of the layer (the caller's cache, tied to the layer's lifetime)
        if (ws is not None and ctx.p == 0 and H == 2 * Ho and W == 2 * Wo and C % 64 == 0 and Cin % 64 == 0
                and tuple(ws.shape) == (4 * Cin, 4 * C) and os.environ.get("DREAMMAT_S2_DGRAD", "subpixel") != "zeroins"):
            if SUBPIXEL_STORE and Cin % 16 == 0:        # the four parities stored where they belong: dx itself, no interleaving copy
                return conv2x2_nhwc(g.contiguous(), ws, None, (Ho, Wo), (1, 1), "conv2x2_dgrad", subpixel=1), None, None, None, None, None
            y = conv2x2_nhwc(g.contiguous(), ws, None, (Ho, Wo), (1, 1), "conv2x2_dgrad")
            dx = y.view(B, Ho, Wo, 2, 2, Cin).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, Cin)       # (u, py, v, px) -> (2u + py, 2v + px)
            return dx, None, None, None, None, None
        g_up = torch.zeros(B, H, W, C, device=g.device, dtype=g.dtype)
        g_up[:, 0:2 * Ho:2, 0:2 * Wo:2] = g
        q = 2 - ctx.p
        return conv3x3_nhwc(g_up, ctx.w_dgrad, None, 1, (q, q), (H, W)), None, None, None, None, None


def conv3x3_s2_autograd(x_nhwc, w_fwd, w_dgrad, bias, lead_pad, w_sub=None):
    """w_sub: subpixel_dgrad_weights(w_fwd[:Cout], Cin) for the data gradient's 2 x 2 form (lead_pad 0), or None."""
    return _Conv3x3S2.apply(x_nhwc, w_fwd, w_dgrad, bias, lead_pad, w_sub)


def conv3x3_train_ok(x_nhwc, weight, stride, padding):
    """the trainable-conv route (MFMA forward, data gradient and weight gradient) serves this layer and input"""
    if not (x_nhwc.is_cuda and x_nhwc.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.shape[2:] == (3, 3)
            and tuple(padding) == (1, 1) and tuple(stride) in ((1, 1), (2, 2))):
        return False
    B, H, W, Cin = x_nhwc.shape
    Cout, s = weight.shape[0], stride[0]
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    if Cin % 64 or Cout % 64 or H % s or W % s or Wo > 64 or Wo & (Wo - 1) or (Ho * Wo) % 64 or Ho % (64 // Wo):
        return False
    rows_in, cols_in = s * (64 // Wo - 1) + 3, s * (Wo - 1) + 3
    return (64 + rows_in * cols_in) * 192 <= 160 * 1024


def conv3x3_wgrad(x_nhwc, dy_nhwc, stride):
    """dW [Cout, Cin, 3, 3] fp32 of a 3x3 / pad 1 conv from its NHWC bf16 input and output gradient (dm_conv3x3_wgrad_nhwc_bf16)."""
    _need_cuda(x_nhwc, dy_nhwc)
    B, H, W, Cin = x_nhwc.shape
    _, Ho, Wo, Cout = dy_nhwc.shape
    splits = int(_lib.lib().dm_conv3x3_wgrad_splits(B, Ho, Wo, Cin, Cout))
    assert splits > 0 and x_nhwc.is_contiguous() and dy_nhwc.is_contiguous()
    part = torch.empty(splits, Cout, 3, 3, Cin, device=x_nhwc.device, dtype=torch.float32)
    with _Timed(f"conv3x3_wgrad[{Cin}->{Cout},{Ho}x{Wo},s{stride}]", 18.0 * B * Ho * Wo * Cin * Cout):
        check(_lib.lib().dm_conv3x3_wgrad_nhwc_bf16(x_nhwc.data_ptr(), dy_nhwc.data_ptr(), part.data_ptr(), B, H, W, Cin, Cout,
                                                    int(stride), _stream()), "dm_conv3x3_wgrad_nhwc_bf16")
    return (part.sum(0) if splits > 1 else part[0]).permute(0, 3, 1, 2)


class _Conv3x3Train(torch.autograd.Function):
    """3x3 / pad 1 conv (stride 1 or 2) with TRAINABLE weights, all three products on the MFMA kernels: forward and data
    gradient = the implicit-GEMM kernel (the data gradient on the flipped, channel-swapped weights; stride 2: on the
    zero-inserted output gradient), weight gradient = dm_conv3x3_wgrad_nhwc_bf16.  No im2col buffer anywhere."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, bias, stride):
        wd = weight.detach()
        Cout, Cin = wd.shape[:2]
        B, H, W, _ = x_nhwc.shape
        ctx.save_for_backward(x_nhwc, wd)
        ctx.stride, ctx.has_bias = stride, bias is not None
        w_fwd = wd.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
        return conv3x3_nhwc(x_nhwc, w_fwd, bias.detach() if bias is not None else None, stride, (1, 1),
                            ((H - 1) // stride + 1, (W - 1) // stride + 1))

    @staticmethod
    def backward(ctx, g):
        x, wd = ctx.saved_tensors
        g = g.contiguous()
        B, H, W, Cin = x.shape
        Cout = wd.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            w_dgrad = wd.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).contiguous()
            if ctx.stride == 1:
                dx = conv3x3_nhwc(g, w_dgrad, None, 1, (1, 1))
            else:
                g_up = torch.zeros(B, H, W, Cout, device=g.device, dtype=g.dtype)
                g_up[:, 0::2, 0::2] = g
                dx = conv3x3_nhwc(g_up, w_dgrad, None, 1, (1, 1), (H, W))
        if ctx.needs_input_grad[1]:
            dw = conv3x3_wgrad(x, g, ctx.stride).to(wd.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g.sum((0, 1, 2), dtype=torch.float32).to(wd.dtype)       # fp32 accumulation without an fp32 copy of g
        return dx, dw, db, None


def conv3x3_train(x_nhwc, weight, bias, stride):
    """y [B,Ho,Wo,Cout] = conv3x3(x) + bias, differentiable wrt x, weight and bias (see conv3x3_train_ok)."""
    return _Conv3x3Train.apply(x_nhwc, weight, bias, int(stride))


# ------------------------------------------------------------------------------------------ group norm
def _gn_fwd(x_nhwc, gamma, beta, eps, act, keep_for_backward=True):
    """keep_for_backward=False: the 2-launch inference entry (coefficients formed inside the apply kernel, nothing saved)."""
    B, H, W, C = x_nhwc.shape
    y = torch.empty_like(x_nhwc)
    ws = torch.empty(int(_lib.lib().dm_groupnorm_workspace_floats(B, C)), device=x_nhwc.device, dtype=torch.float32)
    infer = (not keep_for_backward and C % 8 == 0 and gamma.is_contiguous() and beta.is_contiguous()
             and gamma.data_ptr() % 16 == 0 and beta.data_ptr() % 16 == 0)
    fn, name = _sym("dm_groupnorm_nhwc_infer" if infer else "dm_groupnorm_nhwc_fwd", _same_half(x_nhwc, gamma, beta))
    with _Timed(f"groupnorm_fwd[C={C},HW={H * W}]", 6.0 * B * H * W * C):
        check(fn(x_nhwc.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), ws.data_ptr(), B, H * W, C, float(eps),
                 int(act), _stream()), name)
    return y, ws


class _GroupNormAct(torch.autograd.Function):
    """act(GroupNorm32(x)); with_skip: also hands x back as a second output, so that a caller that uses x twice (norm1 and the
    skip connection of a ResnetBlock2D) receives BOTH gradients here and the backward kernel adds them in its own pass instead
    of autograd in a separate one."""

    @staticmethod
    def forward(ctx, x_nhwc, gamma, beta, eps, act, with_skip=False):
        y, ws = _gn_fwd(x_nhwc, gamma, beta, eps, act)
        ctx.save_for_backward(x_nhwc, gamma, beta, ws)
        ctx.eps, ctx.act, ctx.with_skip = eps, act, with_skip
        ctx.set_materialize_grads(False)       # an unused output's gradient arrives as None, not as a tensor of zeros
        return (y, x_nhwc.view_as(x_nhwc)) if with_skip else y

    @staticmethod
    def backward(ctx, g, g_skip=None):
        x, gamma, beta, ws = ctx.saved_tensors
        B, H, W, C = x.shape
        if g is None:                          # only the skip branch reached the loss
            return g_skip, None, None, None, None, None
        g = g.contiguous()
        if g_skip is not None:
            g_skip = g_skip.contiguous()
        dx = torch.empty_like(x)
        fn, name = _sym("dm_groupnorm_nhwc_bwd_res", _same_half(x, gamma, beta, g, g_skip))
        check(fn(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), g.data_ptr(), g_skip.data_ptr() if g_skip is not None else None,
                 dx.data_ptr(), ws.data_ptr(), B, H * W, C, float(ctx.eps), int(ctx.act), _stream()), name)
        dgamma = dbeta = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # trainable affine parameters (ControlNet training, bf16): per-workgroup channel sums of dz and dz * xhat, added here
            assert x.dtype == torch.bfloat16, "GroupNorm affine gradients: the training loop runs in bf16"
            rows = int(_lib.lib().dm_groupnorm_affine_rows(B, H * W, C))
            cpart = torch.empty(rows, 2, C, device=x.device, dtype=torch.float32)
            check(_lib.lib().dm_groupnorm_nhwc_bwd_affine(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), g.data_ptr(),
                                                          ws.data_ptr(), cpart.data_ptr(), B, H * W, C, float(ctx.eps),
                                                          int(ctx.act), _stream()), "dm_groupnorm_nhwc_bwd_affine")
            sums = cpart.sum(0)
            dbeta, dgamma = sums[0].to(beta.dtype), sums[1].to(gamma.dtype)
        return dx, dgamma, dbeta, None, None, None


# ---- GroupNorm [+ SiLU] with its apply pass folded into the consuming 3 x 3 convolution (round 6, ABI v13)
# tests / tools assign False for the two-call form (apply pass + convolution) it replaces; DREAMMAT_GN_FOLD=0 does the same for a
# whole process (A/B runs of bench.py)
GN_CONV_FOLD = os.environ.get("DREAMMAT_GN_FOLD", "1") != "0"


# The fold pays where the apply pass it removes is expensive: measured (tools/halo_gn_time.py, one box, f16, microseconds: statistics +
# folded convolution | two-launch GroupNorm + the same patch kernel):  8 x 512^2 x 128: 912 | 969    8 x 256^2 x 256: 602 | 690
# 8 x 128^2 x 512: 554 | 562    8 x 64^2 x 512: 155 | 157    24 x 16^2 x 1280: 195 | 179 -- the transform (~110 vector instructions per 1 KB
# piece beside the MFMAs) costs the convolution 40-180 us, the apply pass costs two passes over the tensor.  From 192 MB on (the VAE
# encoder's 512^2 and 256^2 levels at 8 views); tests assign 0 to exercise the fold on small tensors.
GN_FOLD_MIN_BYTES = 192 << 20


def gn_conv3x3_ok(x_nhwc, gamma, cout):
    """the halo-patch kernel serves conv3x3(act(GroupNorm32(x))) at this shape (stride 1, pad 1: dm_conv3x3_gn_ok) and the tensor is
    large enough for the fold to pay"""
    B, H, W, C = x_nhwc.shape
    return (GN_CONV_FOLD and x_nhwc.is_cuda and x_nhwc.dtype in HALF_DTYPES and gamma.dtype == x_nhwc.dtype and C % 64 == 0
            and 2 * B * H * W * C >= GN_FOLD_MIN_BYTES
            and 2 * B * H * W * max(C, cout) <= CONV_MAX_TENSOR_BYTES and bool(_lib.lib().dm_conv3x3_gn_ok(B, H, W, C, cout)))


def _gn_stats(x_nhwc, gamma, beta, eps):
    """statistics + coefficient kernels of GroupNorm32(x): the workspace dm_groupnorm_nhwc_fwd would leave, no output tensor"""
    B, H, W, C = x_nhwc.shape
    ws = torch.empty(int(_lib.lib().dm_groupnorm_workspace_floats(B, C)), device=x_nhwc.device, dtype=torch.float32)
    fn, name = _sym("dm_groupnorm_nhwc_stats", _same_half(x_nhwc, gamma, beta))
    with _Timed(f"groupnorm_stats[C={C},HW={H * W}]", 2.0 * B * H * W * C):
        check(fn(x_nhwc.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), B, H * W, C, float(eps), _stream()), name)
    return ws


def _gn_conv(x_nhwc, ws, act, w_tap_major, bias, rowbias, residual):
    B, H, W, Cin = x_nhwc.shape
    Cout = w_tap_major.shape[0]
    y = torch.empty(B, H, W, Cout, device=x_nhwc.device, dtype=x_nhwc.dtype)
    fn, name = _sym("dm_conv3x3_gn_nhwc_bf16_fused", _same_half(x_nhwc, w_tap_major, bias, rowbias, residual))
    with _Timed(f"conv3x3[gn+{Cin}->{Cout}@{H}x{W},s1]", 2.0 * B * H * W * Cout * 9 * Cin):
        check(fn(x_nhwc.data_ptr(), ws.data_ptr(), int(act), w_tap_major.data_ptr(), bias.data_ptr() if bias is not None else None,
                 rowbias.data_ptr() if rowbias is not None else None, residual.data_ptr() if residual is not None else None,
                 y.data_ptr(), B, H, W, Cin, Cout, _stream()), name)
    return y


class _GnConv3x3S1(torch.autograd.Function):
    """conv3x3(act(GroupNorm32(x))) + bias (+ residual), frozen conv weights, differentiable wrt x (and the residual): _GroupNormAct
    and _Conv3x3S1 as ONE node whose forward never writes the normalised tensor.  Backward = the two nodes' backwards in sequence
    (data-gradient convolution, then the GroupNorm backward kernels on the saved statistics).  with_skip: as _GroupNormAct --
    x is handed back as a second output so that a caller using x twice receives both gradients in the GroupNorm backward pass."""

    @staticmethod
    def forward(ctx, x_nhwc, gamma, beta, eps, act, w_fwd, w_dgrad, bias, residual, with_skip=False):
        ws = _gn_stats(x_nhwc, gamma, beta, eps)
        y = _gn_conv(x_nhwc, ws, act, w_fwd, bias, None, residual)
        ctx.save_for_backward(x_nhwc, gamma, beta, ws)
        ctx.eps, ctx.act, ctx.w_dgrad, ctx.with_skip = eps, act, w_dgrad, with_skip
        ctx.set_materialize_grads(False)
        return (y, x_nhwc.view_as(x_nhwc)) if with_skip else y

    @staticmethod
    def backward(ctx, g, g_skip=None):
        x, gamma, beta, ws = ctx.saved_tensors
        B, H, W, C = x.shape
        d_res = None
        if g is None:
            return g_skip, None, None, None, None, None, None, None, None, None
        g = g.contiguous()
        if ctx.needs_input_grad[8]:
            d_res = g
        dh = conv3x3_nhwc(g, ctx.w_dgrad, None, 1, (1, 1))              # gradient of the (never materialised) normalised tensor
        if g_skip is not None:
            g_skip = g_skip.contiguous()
        dx = torch.empty_like(x)
        fn, name = _sym("dm_groupnorm_nhwc_bwd_res", _same_half(x, gamma, beta, dh, g_skip))
        check(fn(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dh.data_ptr(), g_skip.data_ptr() if g_skip is not None else None,
                 dx.data_ptr(), ws.data_ptr(), B, H * W, C, float(ctx.eps), int(ctx.act), _stream()), name)
        return dx, None, None, None, None, None, None, None, d_res, None


def gn_conv3x3_nhwc(x_nhwc, gamma, beta, eps, act, w_fwd, w_dgrad, bias, rowbias=None, residual=None, with_skip=False):
    """conv3x3(act(GroupNorm32(x)), stride 1, pad 1) + bias (+ rowbias[:, None, None]) (+ residual) without the normalised tensor
    (callers check gn_conv3x3_ok first).  Differentiable wrt x and residual when grad is enabled (frozen gamma / beta / weights; no
    rowbias there); with_skip -> (y, x) as groupnorm_nhwc_skip."""
    _need_cuda(x_nhwc, gamma, beta, w_fwd)
    assert x_nhwc.is_contiguous() and w_fwd.is_contiguous()
    if torch.is_grad_enabled() and (x_nhwc.requires_grad or (residual is not None and residual.requires_grad)):
        assert rowbias is None and not gamma.requires_grad and not beta.requires_grad
        return _GnConv3x3S1.apply(x_nhwc, gamma, beta, eps, act, w_fwd, w_dgrad, bias, residual, with_skip)
    ws = _gn_stats(x_nhwc, gamma, beta, eps)
    y = _gn_conv(x_nhwc, ws, act, w_fwd, bias, rowbias, residual)
    return (y, x_nhwc) if with_skip else y


def groupnorm_nhwc(x_nhwc, gamma, beta, eps, act):
    """x [B,H,W,C] bf16 contiguous -> act(GroupNorm32(x)) [B,H,W,C]; differentiable wrt x, gamma and beta."""
    _need_cuda(x_nhwc, gamma, beta)
    assert x_nhwc.dtype in HALF_DTYPES and x_nhwc.is_contiguous() and gamma.dtype == x_nhwc.dtype
    if torch.is_grad_enabled() and (x_nhwc.requires_grad or gamma.requires_grad or beta.requires_grad):
        return _GroupNormAct.apply(x_nhwc, gamma, beta, eps, act)
    return _gn_fwd(x_nhwc, gamma, beta, eps, act, keep_for_backward=False)[0]


def groupnorm_nhwc_skip(x_nhwc, gamma, beta, eps, act):
    """(act(GroupNorm32(x)), x): the second output IS x, routed through the same autograd node (see _GroupNormAct)."""
    _need_cuda(x_nhwc, gamma, beta)
    assert x_nhwc.dtype in HALF_DTYPES and x_nhwc.is_contiguous() and gamma.dtype == x_nhwc.dtype
    return _GroupNormAct.apply(x_nhwc, gamma, beta, eps, act, True)


# ------------------------------------------------------------------------------------------ ray queries (row f-1)
class MeshBvh:
    """BVH of a fixed mesh (dm_bvh_build, host) with device copies of the node / triangle arrays; the counterpart of
    `RayTracer(vertices, triangles)` (raytracing_renderer.py:20-31)."""

    def __init__(self, v_pos, tri, device=None, grid_res=0):
        v = v_pos.detach().float().cpu().contiguous()
        t = tri.detach().to(torch.int32).cpu().contiguous()
        n_tri = t.shape[0]
        nodes = torch.zeros(2 * n_tri, 8, dtype=torch.int32)                    # 32 B per node
        tris = torch.empty(n_tri, 12, dtype=torch.float32)
        order = torch.empty(n_tri, dtype=torch.int32)
        n_nodes = ctypes.c_int32(0)
        check(_lib.lib().dm_bvh_build(v.data_ptr(), v.shape[0], t.data_ptr(), n_tri, nodes.data_ptr(), tris.data_ptr(),
                                      order.data_ptr(), ctypes.addressof(n_nodes)), "dm_bvh_build")
        self.n_nodes = int(n_nodes.value)
        self.nodes_host, self.tris_host, self.order = nodes[:self.n_nodes].contiguous(), tris, order
        # 4-wide collapse of the same tree (128 B nodes, csrc/bvh_core.h DmBvhNode4): opt-in for the kernels
        nodes4 = torch.zeros(self.n_nodes, 32, dtype=torch.int32)
        n4 = ctypes.c_int32(0)
        check(_lib.lib().dm_bvh_collapse4(self.nodes_host.data_ptr(), self.n_nodes, nodes4.data_ptr(), ctypes.addressof(n4)),
              "dm_bvh_collapse4")
        self.n_nodes4 = int(n4.value)
        self.nodes4_host = nodes4[:self.n_nodes4].contiguous()
        # occupancy grid of the same triangles (csrc/grid_core.h): what the shading kernels walk by default
        self.grid_host = _lib.GridStruct()
        blob, words = ctypes.c_void_p(), ctypes.c_longlong(0)
        check(_lib.lib().dm_grid_build(tris.data_ptr(), n_tri, int(grid_res), ctypes.byref(self.grid_host), ctypes.byref(blob),
                                       ctypes.byref(words)), "dm_grid_build")
        try:
            buf = (ctypes.c_uint32 * words.value).from_address(blob.value)
            self.grid_blob_host = torch.from_numpy(np.frombuffer(buf, dtype=np.int32).copy())
        finally:
            _lib.lib().dm_host_free(blob)
        self.nodes = self.tris = self.grid = self.grid_blob = None
        if device is not None:
            self.nodes, self.tris = self.nodes_host.to(device), self.tris_host.to(device)
            self.grid_blob = self.grid_blob_host.to(device)
            self.grid = self.grid_struct(self.grid_blob)

    def _grid_offsets(self):
        """word offsets of the five 16-byte-padded sections of the blob (dm_grid_build)."""
        h = self.grid_host
        pad4 = lambda w: (w + 3) // 4 * 4
        o_sb = pad4(h.n_words)
        o_off = o_sb + pad4((h.n_words + 63) // 64)
        o_dist = o_off + pad4((h.n_words + 1) // 2)
        n_blocks = ((h.dim[0] + 1) // 2) * ((h.dim[1] + 1) // 2) * ((h.dim[2] + 1) // 2)
        o_occ = o_dist + pad4((n_blocks + 7) // 8)
        o_tri = o_occ + pad4(h.n_occ + 1)
        return 0, o_sb, o_off, o_dist, o_occ, o_tri

    def grid_struct(self, blob):
        """dm_grid whose pointers address the sections of `blob` (bits | sbase | off16 | dist4 | occ_start | cell_tris)."""
        g = _lib.GridStruct()
        ctypes.memmove(ctypes.byref(g), ctypes.byref(self.grid_host), ctypes.sizeof(g))
        base = blob.data_ptr()
        g.bits, g.sbase, g.off16, g.dist4, g.occ_start, g.cell_tris = [base + 4 * o for o in self._grid_offsets()]
        return g

    def grid_sections(self):
        """(bits, rank of every word's first cell, occ_start, triangle id of every record) as host int64 tensors (tests)."""
        h, b = self.grid_host, self.grid_blob_host
        o = self._grid_offsets()
        u = lambda t: t.to(torch.int64) & 0xffffffff
        bits, sbase = u(b[o[0]:o[0] + h.n_words]), u(b[o[1]:o[1] + (h.n_words + 63) // 64])
        off16 = b[o[2]:o[3]].view(torch.int16)[:h.n_words].to(torch.int64) & 0xffff
        rank = sbase.repeat_interleave(64)[:h.n_words] + off16
        d8 = b[o[3]:o[4]].view(torch.uint8).to(torch.int64)
        n_blocks = ((h.dim[0] + 1) // 2) * ((h.dim[1] + 1) // 2) * ((h.dim[2] + 1) // 2)
        dist = torch.stack([d8 & 15, d8 >> 4], -1).reshape(-1)[:n_blocks]
        occ = u(b[o[4]:o[4] + h.n_occ + 1])
        ids = b[o[5]:].reshape(-1, 12)[:, 3].to(torch.int64)
        return bits, rank, occ, ids, dist

    def any_hit(self, origins, dirs, t_max=10.0):
        """hit mask [n] (bool) of rays origins + t*dirs, 0 < t < t_max (miss <=> the reference's depth >= 10)."""
        _need_cuda(origins, dirs)
        if self.nodes is None:
            raise _lib.DmError("MeshBvh was built without a device")
        o, d = _f32c(origins.reshape(-1, 3)), _f32c(dirs.reshape(-1, 3))
        hit = torch.empty(o.shape[0], dtype=torch.uint8, device=o.device)
        check(_lib.lib().dm_bvh_any_hit_rays(self.nodes.data_ptr(), self.tris.data_ptr(), o.data_ptr(), d.data_ptr(),
                                             o.shape[0], float(t_max), hit.data_ptr(), _stream()), "dm_bvh_any_hit_rays")
        return hit.bool().reshape(origins.shape[:-1])

    def any_hit_grid(self, origins, dirs, t_max=10.0):
        """the same query through the occupancy grid (dm_grid_any_hit_rays)."""
        _need_cuda(origins, dirs)
        if self.grid is None:
            raise _lib.DmError("MeshBvh was built without a device")
        o, d = _f32c(origins.reshape(-1, 3)), _f32c(dirs.reshape(-1, 3))
        hit = torch.empty(o.shape[0], dtype=torch.uint8, device=o.device)
        check(_lib.lib().dm_grid_any_hit_rays(ctypes.byref(self.grid), o.data_ptr(), d.data_ptr(), o.shape[0], float(t_max),
                                              hit.data_ptr(), _stream()), "dm_grid_any_hit_rays")
        return hit.bool().reshape(origins.shape[:-1])


def fibonacci_direction_samples(num_samples):
    """the [n, 2] (azimuth, elevation) table in [0,1]^2 that DreamMatMaterial.configure builds from sample_sphere
    (dreammat_material.py:84-98, 389-398): Fibonacci lattice on the upper hemisphere."""
    n = np.arange(num_samples, 2 * num_samples)                                 # begin_elevation = 0: upper half of 2n points
    z = 2.0 * n / (2 * num_samples) - 1.0
    az = 2 * np.pi * n * ((np.sqrt(5) - 1.0) / 2.0) % (2 * np.pi)
    el = np.arcsin(z)
    return torch.from_numpy(np.stack([az * 0.5 / np.pi, 1 - 2 * el / np.pi], -1).astype(np.float32))


class McScene:
    """Everything the Monte-Carlo shading kernels need besides the per-pixel inputs (dm_mc_scene): the mesh BVH, the
    lat-long radiance images of all environments and the two direction tables."""

    def __init__(self, bvh, latlongs, n_diffuse, n_specular, geometry_type="schlick", device=None):
        if geometry_type not in ("schlick", "ggx_smith"):
            raise NotImplementedError(f"geometry_type={geometry_type!r} (dreammat_material.py:606-613)")
        dev = device or bvh.nodes.device
        self.bvh = bvh
        imgs = [torch.as_tensor(x, dtype=torch.float32) for x in latlongs]
        if any(i.shape != imgs[0].shape for i in imgs):
            raise _lib.DmError("all environment lat-long images must share one resolution for the MC shading kernels")
        self.lights = torch.stack(imgs).to(dev).contiguous()                    # [n_env, h, w, 3]
        self.samples_d = fibonacci_direction_samples(n_diffuse).to(dev)
        self.samples_s = fibonacci_direction_samples(n_specular).to(dev)
        self.n_diffuse, self.n_specular = int(n_diffuse), int(n_specular)
        self.hit_words = int(_lib.lib().dm_mc_hit_words(self.n_diffuse, self.n_specular))
        # 4-wide nodes (child boxes in the parent: a quarter of the dependent fetches per ray, same hits by construction) are
        # the default since they were timed (profiles/r02_mc_probe.json); DREAMMAT_BVH=2 selects the binary tree
        self.nodes4 = bvh.nodes4_host.to(dev) if os.environ.get("DREAMMAT_BVH", "4") == "4" else None
        # occlusion queries: the occupancy grid (default since round 3) or, DREAMMAT_MC_TRACER=bvh, the tree
        self.grid = None
        if os.environ.get("DREAMMAT_MC_TRACER", "grid") == "grid":
            self.grid_blob = bvh.grid_blob_host.to(dev)
            self.grid = bvh.grid_struct(self.grid_blob)
        self.struct = _lib.McSceneStruct(bvh.nodes.data_ptr(), bvh.tris.data_ptr(), self.lights.data_ptr(),
                                         self.lights.shape[0], self.lights.shape[1], self.lights.shape[2],
                                         self.samples_d.data_ptr(), self.samples_s.data_ptr(), self.n_diffuse,
                                         self.n_specular, 1 if geometry_type == "ggx_smith" else 0,
                                         self.nodes4.data_ptr() if self.nodes4 is not None else None,
                                         ctypes.addressof(self.grid) if self.grid is not None else None)


class _McShade(torch.autograd.Function):
    """DreamMatMaterial.forward(use_raytracing=True) -> shade_raytracing (dreammat_material.py:615-677, 726-744)."""

    @staticmethod
    def forward(ctx, feat, pos, nrm, view, pix_idx, n_dev, env_of_view, scene, mat, HW, rand_d, rand_s, want_debug):
        _need_cuda(feat, pos, nrm, view, pix_idx, env_of_view, rand_d, rand_s)
        N = feat.shape[0]
        dev = feat.device
        color = torch.empty(3, N, device=dev)
        hit_bits = torch.zeros(max(N, 1), scene.hit_words, dtype=torch.int32, device=dev)
        dbg = [None] * 7
        if want_debug:
            dbg = [torch.empty(N, 3, device=dev) for _ in range(5)] + [torch.empty(N, 1, device=dev) for _ in range(2)]
        if N > 0:
            with _Timed("mc_shade_fwd", 72.0 * N):
                check(_lib.lib().dm_mc_shade_fwd(
                    ctypes.byref(scene.struct), ctypes.byref(mat), pos.data_ptr(), *_rs_cs(pos), nrm.data_ptr(), *_rs_cs(nrm),
                    view.data_ptr(), *_rs_cs(view), feat.data_ptr(), *_rs_cs(feat), pix_idx.data_ptr(), env_of_view.data_ptr(),
                    n_dev.data_ptr(), N, HW, rand_d.data_ptr() if rand_d is not None else None,
                    rand_s.data_ptr() if rand_s is not None else None, hit_bits.data_ptr(), color.data_ptr(), 1, N,
                    *[d.data_ptr() if d is not None else None for d in dbg], _stream()), "dm_mc_shade_fwd")
        ctx.save_for_backward(feat, pos, nrm, view, pix_idx, n_dev, env_of_view, hit_bits,
                              *([rand_d] if rand_d is not None else []), *([rand_s] if rand_s is not None else []))
        ctx.has_rand = (rand_d is not None, rand_s is not None)
        ctx.scene, ctx.mat, ctx.HW = scene, mat, HW
        # order of the debug buffers: albedo, specular_lights, diffuse_lights, specular_colors, diffuse_colors,
        # metalness, roughness (as dm_shade_fwd)
        outs = (color.t(),) + tuple(d for d in dbg if d is not None)
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g, *unused):
        saved = list(ctx.saved_tensors)
        feat, pos, nrm, view, pix_idx, n_dev, env_of_view, hit_bits = saved[:8]
        rest = saved[8:]
        rand_d = rest.pop(0) if ctx.has_rand[0] else None
        rand_s = rest.pop(0) if ctx.has_rand[1] else None
        N = feat.shape[0]
        dfeat = torch.zeros(5, N, device=feat.device)
        if N > 0:
            with _Timed("mc_shade_bwd", 92.0 * N):
                check(_lib.lib().dm_mc_shade_bwd(
                    ctypes.byref(ctx.scene.struct), ctypes.byref(ctx.mat), pos.data_ptr(), *_rs_cs(pos), nrm.data_ptr(),
                    *_rs_cs(nrm), view.data_ptr(), *_rs_cs(view), feat.data_ptr(), *_rs_cs(feat), pix_idx.data_ptr(),
                    env_of_view.data_ptr(), n_dev.data_ptr(), N, ctx.HW, rand_d.data_ptr() if rand_d is not None else None,
                    rand_s.data_ptr() if rand_s is not None else None, hit_bits.data_ptr(), g.data_ptr(), *_rs_cs(g),
                    dfeat.data_ptr(), 1, N, _stream()), "dm_mc_shade_bwd")
        return (dfeat.t(),) + (None,) * 12


def mc_shade(feat, pos, nrm, view, pix_idx, n_dev, env_of_view, scene, mat, HW, rand_d=None, rand_s=None, want_debug=True):
    """-> (color [N,3], albedo, spec_light, diff_light, spec_color, diff_color, metallic, roughness), the 8 outputs of
    shade_raytracing (all already lin2srgb-encoded where the reference encodes them)."""
    return _McShade.apply(feat, pos, nrm, view, pix_idx, n_dev, env_of_view, scene, mat, HW, rand_d, rand_s, want_debug)


# ------------------------------------------------------------------------------------------ transformer rows
def layernorm_rows(x, gamma, beta, eps):
    """LayerNorm over the last dim of a contiguous bf16 tensor [..., C] (forward only)."""
    _need_cuda(x, gamma, beta)
    assert x.is_contiguous()
    fn, name = _sym("dm_layernorm_bf16", _same_half(x, gamma, beta))
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    with _Timed(f"layernorm[C={C}]", 4.0 * rows * C):
        check(fn(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), rows, C, float(eps), _stream()), name)
    return y


class _SoftmaxRows(torch.autograd.Function):
    """p = softmax(scale * s) over the last dim of a bf16 score tensor, fp32 arithmetic, with its backward (the differentiated
    VAE mid-block attention, sd/models.py VaeAttention; dm_softmax_rows_bf16 / dm_softmax_rows_bwd_bf16)."""

    @staticmethod
    def forward(ctx, s, scale):
        s = s.contiguous()
        cols = s.shape[-1]
        rows = s.numel() // cols
        p = torch.empty_like(s)
        with _Timed(f"softmax_rows[{cols}]", 4.0 * rows * cols):
            fn, name = _sym("dm_softmax_rows_bf16", s.dtype)
            check(fn(s.data_ptr(), p.data_ptr(), rows, cols, float(scale), _stream()), name)
        ctx.save_for_backward(p)
        ctx.scale = float(scale)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        dp = dp.to(p.dtype).contiguous()
        cols = p.shape[-1]
        rows = p.numel() // cols
        ds = torch.empty_like(p)
        with _Timed(f"softmax_rows_bwd[{cols}]", 6.0 * rows * cols):
            fn, name = _sym("dm_softmax_rows_bwd_bf16", p.dtype)
            check(fn(p.data_ptr(), dp.data_ptr(), ds.data_ptr(), rows, cols, ctx.scale, _stream()), name)
        return ds, None


def transpose_rows(x, out=None):
    """x [..., R, C] (16-bit, contiguous) -> [..., C, R] contiguous (dm_transpose_bf16 / _f16; R % 64 == C % 64 == 0)."""
    _need_cuda(x)
    assert x.dtype in HALF_DTYPES and x.is_contiguous() and x.dim() >= 2
    R, C = x.shape[-2], x.shape[-1]
    batch = x.numel() // (R * C)
    if out is None:
        out = torch.empty(*x.shape[:-2], C, R, device=x.device, dtype=x.dtype)
    assert out.is_contiguous() and out.dtype == x.dtype and out.numel() == x.numel() and out.data_ptr() != x.data_ptr()
    with _Timed(f"transpose[{R}x{C}]", 4.0 * x.numel()):
        fn, name = _sym("dm_transpose_bf16", x.dtype)
        check(fn(x.data_ptr(), out.data_ptr(), batch, R, C, _stream()), name)
    return out


def _softmax_rows_(s, scale):
    """in place: s <- softmax(scale * s) per row (no autograd)"""
    cols = s.shape[-1]
    rows = s.numel() // cols
    with _Timed(f"softmax_rows[{cols}]", 4.0 * rows * cols):
        fn, name = _sym("dm_softmax_rows_bf16", s.dtype)
        check(fn(s.data_ptr(), s.data_ptr(), rows, cols, float(scale), _stream()), name)
    return s


def _softmax_rows_bwd_(p, dp, scale):
    """in place: dp <- scale * p * (dp - rowsum(p * dp)) (no autograd)"""
    cols = p.shape[-1]
    rows = p.numel() // cols
    with _Timed(f"softmax_rows_bwd[{cols}]", 6.0 * rows * cols):
        fn, name = _sym("dm_softmax_rows_bwd_bf16", p.dtype)
        check(fn(p.data_ptr(), dp.data_ptr(), dp.data_ptr(), rows, cols, float(scale), _stream()), name)
    return dp


_WIDE_ATTN_WS = {}      # (device, dtype, G, Sq, Skv, D) -> workspace of wide_head_attention, reused by every call (one stream)
WIDE_ATTN_GROUP_BYTES = 128 << 20      # score bytes (G * Sq * Skv * 2) one launch group may cover: 4 images at S = 4096, 1 at 16384


def _wide_attn_group(B, Sq, Skv):
    """images per launch: enough to fill the chip's 256 CUs with tiles (one 4096 x 4096 x 512 product is ONE round of 256 tiles, a
    [4096, 512] output 64 tiles), bounded so the score workspace stays independent of the batch"""
    g = 1
    while g * 2 <= B and B % (g * 2) == 0 and (g * 2) * Sq * Skv * 2 <= WIDE_ATTN_GROUP_BYTES:
        g *= 2
    return g


def _wide_attn_ws(q, G, Sq, Skv, D, backward):
    """score-sized buffers ([G, Sq, Skv]: one forward, four backward) and [G, D, S] buffers for the transposed operands of a group"""
    key = (q.device, q.dtype, G, Sq, Skv, D)
    ws = _WIDE_ATTN_WS.get(key)
    if ws is None:
        ws = _WIDE_ATTN_WS[key] = {"score": [], "small": []}
    mk = lambda n: torch.empty(n, device=q.device, dtype=q.dtype)
    while len(ws["score"]) < (4 if backward else 1):
        ws["score"].append(mk(G * Sq * Skv))
    while len(ws["small"]) < (3 if backward else 1):
        ws["small"].append(mk(G * D * max(Sq, Skv)))
    return ws["score"], ws["small"]


def wide_head_attention_ok(q, k, v):
    """q [B, Sq, D], k / v [B, Skv, D]: every product of the GEMM form (forward and backward) inside dm_gemm_*_batched's and
    dm_transpose_*'s domains, the rows inside dm_softmax_rows_*'s."""
    if not (q.is_cuda and q.dtype in HALF_DTYPES and q.dim() == 3 and k.shape == v.shape and k.shape[0] == q.shape[0] and k.shape[2] == q.shape[2]):
        return False
    Sq, D, Skv = q.shape[1], q.shape[2], k.shape[1]
    return Sq % 256 == 0 and Skv % 256 == 0 and D % 64 == 0 and Skv <= 16384 and 2 * Sq * Skv <= CONV_MAX_TENSOR_BYTES // 4


class _WideHeadAttention(torch.autograd.Function):
    """softmax(scale * q k^T) v for ONE head too wide for the flash kernels' register tiles (AutoencoderKL's mid block: d = 512,
    S = 4096 at 512^2; the one differentiated attention of the path, dreammat_guidance.py:284-292): a few images at a time on the
    hand-written GEMM (dm_gemm_*_batched), the row-softmax kernels and dm_transpose_*, over ONE reused [G, Sq, Skv] score buffer
    (four in the backward; G = 4 images at S = 4096, whatever the batch) -- no [B, S, S] tensor in either direction, nothing saved
    but q, k, v: the backward recomputes a group's probabilities (one more product and softmax pass).  Products per group:
    forward s = q k^T, o = p (v^T)^T; backward s, dp = do v^T, dq = ds (k^T)^T, dv = p^T (do^T)^T, dk = ds^T (q^T)^T -- every operand
    contracted along its rows is transposed first (v, k, q, do of the group into [G, D, S] buffers; p, ds into two more score
    buffers)."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, Sq, D = q.shape
        Skv = k.shape[1]
        G = _wide_attn_group(B, Sq, Skv)
        score, small = _wide_attn_ws(q, G, Sq, Skv, D, False)
        s, vt = score[0].view(G, Sq, Skv), small[0][:G * D * Skv].view(G, D, Skv)
        o = torch.empty_like(q)
        for b in range(0, B, G):
            gemm_batched(q[b:b + G], k[b:b + G], s)
            _softmax_rows_(s, scale)
            transpose_rows(v[b:b + G], out=vt)
            gemm_batched(s, vt, o[b:b + G])
        ctx.save_for_backward(q, k, v)
        ctx.scale = float(scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v = ctx.saved_tensors
        B, Sq, D = q.shape
        Skv = k.shape[1]
        do = do.contiguous()
        G = _wide_attn_group(B, Sq, Skv)
        score, small = _wide_attn_ws(q, G, Sq, Skv, D, True)
        p, dp = score[0].view(G, Sq, Skv), score[1].view(G, Sq, Skv)
        pt, dst = score[2].view(G, Skv, Sq), score[3].view(G, Skv, Sq)
        kt, qt, dot = small[0][:G * D * Skv].view(G, D, Skv), small[1][:G * D * Sq].view(G, D, Sq), small[2][:G * D * Sq].view(G, D, Sq)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        for b in range(0, B, G):
            g = slice(b, b + G)
            gemm_batched(q[g], k[g], p)
            _softmax_rows_(p, ctx.scale)
            gemm_batched(do[g], v[g], dp)
            _softmax_rows_bwd_(p, dp, ctx.scale)             # dp <- ds
            transpose_rows(k[g], out=kt)
            gemm_batched(dp, kt, dq[g])
            transpose_rows(p, out=pt)
            transpose_rows(do[g], out=dot)
            gemm_batched(pt, dot, dv[g])
            transpose_rows(dp, out=dst)
            transpose_rows(q[g], out=qt)
            gemm_batched(dst, qt, dk[g])
        return dq, dk, dv, None


def wide_head_attention(q, k, v, scale):
    """softmax(scale * q k^T) v, q [B, Sq, D], k / v [B, Skv, D] 16-bit (one head), differentiable; see _WideHeadAttention.
    The caller keeps scale * |q k^T| inside the element type's range (IEEE half: fold a power of two into q)."""
    _need_cuda(q, k, v)
    assert wide_head_attention_ok(q, k, v)
    return _WideHeadAttention.apply(q, k, v, scale)


def softmax_rows_ok(s):
    return s.is_cuda and s.dtype in HALF_DTYPES and s.shape[-1] % 8 == 0 and s.shape[-1] <= 16384


def softmax_rows(s, scale):
    """softmax(scale * s, dim=-1) of a bf16 tensor in one pass each way (differentiable)."""
    _need_cuda(s)
    assert softmax_rows_ok(s)
    return _SoftmaxRows.apply(s, scale)


def geglu_rows(h):
    """h [..., 2*inner] bf16 contiguous -> h[..., :inner] * gelu(h[..., inner:]) (forward only)."""
    _need_cuda(h)
    assert h.dtype in HALF_DTYPES and h.is_contiguous() and h.shape[-1] % 2 == 0
    inner = h.shape[-1] // 2
    rows = h.numel() // (2 * inner)
    y = torch.empty(*h.shape[:-1], inner, device=h.device, dtype=h.dtype)
    with _Timed(f"geglu[inner={inner}]", 6.0 * rows * inner):
        fn, name = _sym("dm_geglu_bf16", h.dtype)
        check(fn(h.data_ptr(), y.data_ptr(), rows, inner, _stream()), name)
    return y


def cat_add_nhwc(x, s, r=None, r_scale=1.0):
    """x [B,Cx,H,W], s / r [B,Cs,H,W] bf16 in channels-last memory -> cat([x, s + r_scale * r], dim=1), channels-last
    (forward only): the up-block skip connection with the ControlNet residual folded in, one pass."""
    _need_cuda(x, s)
    B, Cx, H, W = x.shape
    Cs = s.shape[1]
    xn, sn = x.permute(0, 2, 3, 1), s.permute(0, 2, 3, 1)
    rn = r.permute(0, 2, 3, 1) if r is not None else None
    assert xn.is_contiguous() and sn.is_contiguous() and (rn is None or rn.is_contiguous())
    fn, name = _sym("dm_cat_add_bf16", _same_half(x, s, r))
    y = torch.empty(B, H, W, Cx + Cs, device=x.device, dtype=x.dtype)
    with _Timed(f"cat_add[Cx={Cx},Cs={Cs}]", 2.0 * B * H * W * (2 * Cx + (3 if r is not None else 2) * Cs)):
        check(fn(xn.data_ptr(), sn.data_ptr(), rn.data_ptr() if rn is not None else None, y.data_ptr(), B * H * W, Cx, Cs,
                 float(r_scale), _stream()), name)
    return y.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------ optimiser
def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, grad_scale=1.0, zero_grad=True):
    _need_cuda(param, grad, exp_avg, exp_avg_sq)
    check(_lib.lib().dm_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                  param.numel(), int(step), float(lr), float(beta1), float(beta2), float(eps),
                                  float(grad_scale), int(bool(zero_grad)), _stream()), "dm_adam_step")
