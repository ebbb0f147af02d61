"""Building blocks of the Stable-Diffusion nets (UNet2DConditionModel / ControlNetModel / AutoencoderKL
encoder) with diffusers-compatible parameter names, so `from_pretrained` checkpoints of
`stabilityai/stable-diffusion-2-1-base` (dreammat.yaml:60), `runwayml/stable-diffusion-v1-5` and the
22-channel `zzzyuqing/light-geo-controlnet` (README.md:26) load with `load_state_dict(strict=True)`.

diffusers is an un-vendored dependency of the reference (requirements.txt:7); it is used through
threestudio/models/guidance/dreammat_guidance.py:110-154, 205-292.  On the GPU in bf16 every layer here runs in the
hand-written gfx950 kernels of csrc/ (DESIGN.md sections 1 and 3): 3x3 convs and their data gradients in the persistent
implicit-GEMM kernel (conv.hip / conv_small.hip), Linear / 1x1 layers with bias / residual / GEGLU epilogues in its 1-tap
instantiation, GroupNorm(+SiLU) fwd / bwd (groupnorm.hip), LayerNorm / GEGLU / the skip concat (transformer.hip), the
QK^T.softmax.V of every transformer block in attn_w64.hip / attention.hip.  What stays on ATen / hipBLASLt is listed in
DESIGN.md section 1 (the VAE mid-block attention's projections, anything differentiated that has no hand-written backward).
fp32 / CPU execution (BASELINE config 1: fp32 plumbing run, the fp32 legs of the parity tests) uses the plain torch paths
below.  A CUDA 16-bit call that leaves the hand-written kernels (a shape outside their gates, a differentiated layer without a
hand-written backward) is RECORDED and announced once per (layer kind, shape) on stderr -- `fallbacks()` returns the record,
`DREAMMAT_STRICT_KERNELS=1` turns the first such call into an error (tests pin the set a step is allowed to contain).
"""
import math
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hipops


# The product runs ONE lowering on the GPU: "mfma" -- 3x3 convs in the hand-written implicit-GEMM MFMA kernels (csrc/conv.hip,
# csrc/conv_small.hip) on NHWC activations, Linear / 1x1 layers on the 1-tap instantiation of the same kernel.  There is no
# environment switch.  Tests and tools/*_probe.py may assign `layers.CONV_BACKEND = "gemm"` (im2col + hipBLASLt, ATen
# elementwise: the A/B baseline and the fp32 evaluation the bf16 kernels are compared with) for the duration of a measurement;
# "miopen" = torch's default conv, unusable in this image (no gfx950 kernel database: every shape JIT-compiles, ~25 min).
# Tensors that are not on the GPU (the CPU test tier: fp32 plumbing of the same modules) take the ATen path.
CONV_BACKEND = "mfma"

HALF = (torch.bfloat16, torch.float16)       # element types the net kernels are built for (csrc/dm_elem.h; the reference's nets
#                                               run in fp16, dreammat_guidance.py:56,92-94; bf16 is BASELINE's 1-GPU configuration)
_FALLBACKS = {}


def note_fallback(kind, detail):
    """a CUDA bf16 / f16 call is about to run on ATen / hipBLASLt instead of the kernels of csrc/: record it, say so once."""
    key = (kind, detail)
    n = _FALLBACKS.get(key, 0)
    _FALLBACKS[key] = n + 1
    if os.environ.get("DREAMMAT_STRICT_KERNELS") == "1":
        raise RuntimeError(f"[dreammat_amd] {kind} {detail}: no hand-written kernel for this call (DREAMMAT_STRICT_KERNELS=1)")
    if n == 0:
        print(f"[dreammat_amd] {kind} {detail}: outside the hand-written kernels, running on ATen / hipBLASLt", file=sys.stderr)


def fallbacks(clear=False):
    """{(layer kind, shape / reason): calls} of every 16-bit CUDA call that left the hand-written kernels in this process"""
    out = dict(_FALLBACKS)
    if clear:
        _FALLBACKS.clear()
    return out


def _native_expected(x):
    """the product lowering is selected and the tensor is one the kernels of csrc/ exist for"""
    return CONV_BACKEND == "mfma" and x.is_cuda and x.dtype in HALF
# Differentiated layers with TRAINABLE parameters (the ControlNet training loop, row f-4): MFMA attention forward + backward,
# 3x3 convolutions with the weight-gradient kernel, GroupNorm with affine gradients.  tools/train_step_probe.py assigns False
# to time the torch-autograd lowering the kernels replace (im2col + hipBLASLt, matmul-softmax, ATen GroupNorm); no
# environment switch.
TRAIN_KERNELS = True
# V^T projections of the self-attention layers through the fused GEMM kernel with swapped operands (project_vt); tools may
# assign False to time the hipBLASLt strided-batched product it replaces.
VT_BY_FUSED_GEMM = True
# q | k of the frozen self-attention layers from one GEMM (Attention._qk_fused_weight); tests assign False for the two-GEMM reference
QK_FUSED = True
# "fp8": the self-attention of the frozen nets (64-wide heads, S >= 1024) on the MX-FP8 matrix instruction (csrc/attn_fp8.hip) --
# BASELINE configs[4] "fp8 MFMA attention".  "16bit" (default): the bf16 / f16 kernels.  The setting is carried by every
# Attention MODULE (`attention_precision`, set on a guidance's own nets by set_attention_precision below -- two guidance objects
# in one process do not switch each other); this module attribute is only the default of modules nobody configured
# (tests assign it).
ATTENTION_PRECISION = "16bit"
FP8_MIN_SEQ = 1024


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict keys) with JIT-free GPU lowerings."""

    def _prepared(self):
        w = self.weight
        key = (w.data_ptr(), w._version, w.dtype)
        if getattr(self, "_prep_key", None) != key:
            Cout, Cin = w.shape[0], w.shape[1]
            wd, bd = w.detach(), (self.bias.detach() if self.bias is not None else None)
            if Cout % 64:
                # narrow heads (UNet conv_out 320->4, VAE conv_out 512->8) and the 96-wide stem layers: zero-pad Cout to
                # whole 64-wide MFMA tiles instead of an im2col + GEMM lowering; callers slice the first Cout channels back out
                padc = -Cout % 64
                wd = torch.cat([wd, wd.new_zeros(padc, Cin, 3, 3)], dim=0)
                bd = torch.cat([bd, bd.new_zeros(padc)]) if bd is not None else None
            Cp = wd.shape[0]
            self._w_fwd = wd.permute(0, 2, 3, 1).reshape(Cp, 9 * Cin).contiguous()
            self._w_dgrad = wd.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9 * Cp).contiguous()
            self._bias_p = bd
            self._prep_key = key
        return self._w_fwd, self._w_dgrad

    def _forward_gemm(self, x):
        B, Cin, H, W = x.shape
        Cout, _, kh, kw = self.weight.shape
        sh, sw = self.stride
        ph, pw = self.padding
        if kh == 1 and kw == 1 and sh == 1 and sw == 1 and ph == 0 and pw == 0:
            xn = x.permute(0, 2, 3, 1)                                   # free for channels-last activations
            y = linear_fused(xn, self.weight.view(Cout, Cin), self.bias)  # [B,H,W,Cout]
            return y.permute(0, 3, 1, 2)
        cols = F.unfold(x, (kh, kw), padding=(ph, pw), stride=(sh, sw))  # [B, Cin*kh*kw, L]
        # [B, L, K] @ [K, Cout]: the result is produced directly in NHWC, like every other layer of the nets
        y = torch.matmul(cols.transpose(1, 2), self.weight.view(Cout, Cin * kh * kw).t())
        if self.bias is not None:
            y = y + self.bias
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        return y.view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)

    def _frozen(self):
        return not (torch.is_grad_enabled() and self.weight.requires_grad)

    def mfma_ok(self, x):
        Cout, Cin, kh, kw = self.weight.shape
        return (CONV_BACKEND == "mfma" and x.is_cuda and x.dtype in HALF and kh == 3 and kw == 3
                and Cin % 32 == 0 and Cout % 64 == 0 and self._frozen())

    def fused_ok(self, x):
        """conv + per-image channel bias + residual in one kernel: stride-1 LDS-DMA path, nothing needs autograd."""
        return (self.mfma_ok(x) and self.weight.shape[1] % 64 == 0 and self.stride == (1, 1) and self.padding == (1, 1)
                and not (torch.is_grad_enabled() and x.requires_grad))

    def forward_fused(self, x, rowbias=None, residual=None):
        """y = conv(x) + bias + rowbias[:, :, None, None] + residual (logical NCHW in / out, NHWC in memory)."""
        w_fwd, _ = self._prepared()
        xn = x.permute(0, 2, 3, 1).contiguous()
        rn = residual.permute(0, 2, 3, 1).contiguous() if residual is not None else None
        rb = rowbias.contiguous() if rowbias is not None else None
        return hipops.conv3x3_nhwc(xn, w_fwd, self._bias_p, 1, (1, 1), None, rb, rn).permute(0, 3, 1, 2)

    def forward_residual(self, x, residual):
        """conv(x) + residual; the add rides in the kernel epilogue on the stride-1 LDS-DMA path (also under
        autograd: the VAE encoder's ResnetBlock2D), otherwise a separate add."""
        Cout, Cin = self.weight.shape[:2]
        if not (self.mfma_ok(x) and Cin % 64 == 0 and self.stride == (1, 1) and self.padding == (1, 1)
                and residual.dtype == x.dtype):
            return self.forward(x) + residual
        w_fwd, w_dgrad = self._prepared()
        xn = x.permute(0, 2, 3, 1).contiguous()
        rn = residual.permute(0, 2, 3, 1).contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or residual.requires_grad):
            y = hipops.conv3x3_s1_autograd(xn, w_fwd, w_dgrad, self._bias_p, rn)
        else:
            y = hipops.conv3x3_nhwc(xn, w_fwd, self._bias_p, 1, (1, 1), None, None, rn)
        return y.permute(0, 3, 1, 2)

    def gn_fold_ok(self, norm, x):
        """conv(act(norm(x))) as ONE convolution launch with the GroupNorm apply pass folded in (hipops.gn_conv3x3_nhwc, round 6):
        frozen 3 x 3 stride-1 pad-1 layer, frozen 32-group norm, a shape the halo-patch kernel serves."""
        Cout, Cin = self.weight.shape[:2]
        return (self.mfma_ok(x) and Cin % 64 == 0 and Cout % 64 == 0 and self.stride == (1, 1) and self.padding == (1, 1)
                and _gn_kernel_ok(norm, x) and not (torch.is_grad_enabled() and (norm.weight.requires_grad or norm.bias.requires_grad))
                and hipops.gn_conv3x3_ok(x.permute(0, 2, 3, 1), norm.weight, Cout))

    def forward_gn_fold(self, norm, x, silu=True, rowbias=None, residual=None, with_skip=False):
        """conv(act(norm(x))) + bias (+ rowbias[:, :, None, None]) (+ residual); logical NCHW in / out.  with_skip -> (y, x): x routed
        through the same autograd node (a ResnetBlock2D's input feeds norm1 and the skip connection)."""
        w_fwd, w_dgrad = self._prepared()
        xn = x.permute(0, 2, 3, 1).contiguous()
        rn = residual.permute(0, 2, 3, 1).contiguous() if residual is not None else None
        rb = rowbias.contiguous() if rowbias is not None else None
        out = hipops.gn_conv3x3_nhwc(xn, norm.weight, norm.bias, norm.eps, 1 if silu else 0, w_fwd, w_dgrad, self._bias_p, rb, rn, with_skip)
        if with_skip:
            return out[0].permute(0, 3, 1, 2), out[1].permute(0, 3, 1, 2)
        return out.permute(0, 3, 1, 2)

    def small_ok(self, x):
        """few-channel stem layers on the direct kernel (forward only, frozen nets)."""
        Cout, Cin, kh, kw = self.weight.shape
        return (CONV_BACKEND == "mfma" and x.is_cuda and x.dtype in HALF and kh == 3 and kw == 3
                and Cin in hipops.SMALL_CONV_CIN and Cout % 16 == 0 and self.stride[0] == self.stride[1]
                and self.padding[0] == self.padding[1] and self._frozen() and not (torch.is_grad_enabled() and x.requires_grad))

    def forward_small(self, x, act, residual=None):
        """act(conv(x) + bias (+ residual)) with act = 0 none / 1 SiLU fused before the rounding (ControlNetConditioningEmbedding);
        residual: logical [Br, Cout, Ho, Wo] with B % Br == 0, image b takes residual image b % Br."""
        w_fwd, _ = self._prepared()
        Cout = self.weight.shape[0]
        xn = x.permute(0, 2, 3, 1).contiguous()
        rn = residual.permute(0, 2, 3, 1).contiguous() if residual is not None else None
        y = hipops.conv3x3_small_nhwc(xn, w_fwd[:Cout], self._bias_p[:Cout] if self._bias_p is not None else None,
                                      self.stride[0], tuple(self.padding), act, rn)
        return y.permute(0, 3, 1, 2)

    def forward_silu(self, x):
        """F.silu(self(x)); one kernel on the few-channel stem layers."""
        return self.forward_small(x, 1) if self.small_ok(x) else F.silu(self(x))

    def forward_strided_asym(self, x):
        """stride-2 conv over F.pad(x, (0,1,0,1)) (AutoencoderKL downsampler) without materialising the pad."""
        w_fwd, w_dgrad = self._prepared()
        xn = x.permute(0, 2, 3, 1).contiguous()
        if torch.is_grad_enabled() and x.requires_grad:
            Cout, Cin = self.weight.shape[:2]
            key = (self.weight.data_ptr(), self.weight._version, self.weight.dtype)
            if getattr(self, "_sub_key", None) != key:           # sub-pixel weights of the data gradient, cached on the layer
                with torch.no_grad():
                    self._w_sub = hipops.subpixel_dgrad_weights(w_fwd[:Cout], Cin) if Cin % 64 == 0 and Cout % 64 == 0 else None
                self._sub_key = key
            y = hipops.conv3x3_s2_autograd(xn, w_fwd, w_dgrad, self._bias_p, 0, self._w_sub)
        else:
            H, W = x.shape[2], x.shape[3]
            y = hipops.conv3x3_nhwc(xn, w_fwd, self._bias_p, 2, (0, 0), (H // 2, W // 2))
        return y.permute(0, 3, 1, 2)

    def _stem_prepared(self):
        """weights of a <= 4-channel stem layer for the stem kernel (input padded to 4 channels) and for its data gradient."""
        w = self.weight
        key = (w.data_ptr(), w._version, w.dtype)
        if getattr(self, "_stem_key", None) != key:
            Cout, Cin = w.shape[:2]
            wd = w.detach()
            w4 = wd.new_zeros(Cout, 3, 3, 4)
            w4[..., :Cin] = wd.permute(0, 2, 3, 1)
            self._w_stem4 = w4.reshape(Cout, 36).contiguous()
            wg = wd.new_zeros(4, 3, 3, Cout)                       # data gradient = the same conv on the flipped weights, Cout -> Cin (padded to 4)
            wg[:Cin] = wd.flip(2, 3).permute(1, 2, 3, 0)
            self._w_stem_dgrad4 = wg.reshape(4, 9 * Cout).contiguous()
            self._stem_key = key
        return self._w_stem4, self._w_stem_dgrad4

    def forward(self, x):
        if CONV_BACKEND == "miopen" or not x.is_cuda:
            return super().forward(x)
        Cout, Cin, kh, kw = self.weight.shape
        needs_grad = torch.is_grad_enabled() and x.requires_grad
        if self.small_ok(x):
            return self.forward_small(x, 0)
        if (needs_grad and CONV_BACKEND == "mfma" and x.dtype in HALF and kh == 3 and kw == 3 and Cin <= 4
                and Cout % 8 == 0 and (Cout <= 32 or Cout == 128) and self.stride == (1, 1) and self.padding == (1, 1) and self._frozen()):
            # the VAE encoder's conv_in on the rendered image (the only few-channel layer that needs a gradient)
            w4, w_dgrad4 = self._stem_prepared()
            bias = self.bias.detach() if self.bias is not None else None
            return hipops.conv3x3_stem_autograd(x.permute(0, 2, 3, 1), w4, w_dgrad4, bias).permute(0, 3, 1, 2)
        # zero-padded to whole 64-wide tiles (_prepared): always for the LDS-DMA kernel's shapes, and for the register-staged
        # kernel (Cin % 64 != 0) when nothing needs a gradient
        narrow = Cout % 64 != 0 and (Cin % 64 == 0 or (Cin % 32 == 0 and not needs_grad))
        ok = (CONV_BACKEND == "mfma" and x.dtype in HALF and kh == 3 and kw == 3 and Cin % 32 == 0
              and (Cout % 64 == 0 or narrow) and self.stride[0] == self.stride[1] and self._frozen()
              and (not needs_grad or (self.stride[0] == 1 and self.padding == (1, 1) and Cin % 64 == 0)))
        if not ok:
            if CONV_BACKEND == "mfma" and TRAIN_KERNELS and not self._frozen():
                xn = x.permute(0, 2, 3, 1).contiguous()
                if hipops.conv3x3_train_ok(xn, self.weight, self.stride, self.padding):
                    # trainable layer (ControlNet training): forward, data gradient and weight gradient on the MFMA kernels
                    return hipops.conv3x3_train(xn, self.weight, self.bias, self.stride[0]).permute(0, 3, 1, 2)
            if _native_expected(x) and not (kh == 1 and kw == 1):      # (1 x 1 layers: linear_fused below records its own exits)
                note_fallback("conv", f"{kh}x{kw} {Cin}->{Cout} s{self.stride[0]} p{self.padding[0]}"
                                      f"{' autograd' if needs_grad else ''}{'' if self._frozen() else ' trainable'}")
            return self._forward_gemm(x)
        w_fwd, w_dgrad = self._prepared()
        bias = self._bias_p
        xn = x.permute(0, 2, 3, 1).contiguous()                          # no-op when already NHWC
        if needs_grad:
            y = hipops.conv3x3_s1_autograd(xn, w_fwd, w_dgrad, bias)
        else:
            H, W = x.shape[2], x.shape[3]
            s = self.stride[0]
            ph, pw = self.padding
            y = hipops.conv3x3_nhwc(xn, w_fwd, bias, s, (ph, pw), ((H + 2 * ph - 3) // s + 1, (W + 2 * pw - 3) // s + 1))
        if narrow:
            y = y[..., :Cout]
        return y.permute(0, 3, 1, 2)


def _gn_kernel_ok(norm: nn.GroupNorm, x):
    return (x.is_cuda and x.dtype in HALF and norm.num_groups == 32 and CONV_BACKEND == "mfma"
            and norm.weight.dtype == x.dtype
            # trainable affine parameters (ControlNet training): the affine-gradient kernel exists for bf16 only
            and (not (torch.is_grad_enabled() and norm.weight.requires_grad) or (TRAIN_KERNELS and x.dtype == torch.bfloat16)))


def group_norm_act(norm: nn.GroupNorm, x, silu: bool):
    """act(GroupNorm(x)) for a logical-NCHW tensor.  GPU bf16 (32 groups) -> fused NHWC HIP kernel (the
    activation stays channels-last, which is what the implicit-GEMM conv consumes); otherwise torch."""
    if _gn_kernel_ok(norm, x):
        xn = x.permute(0, 2, 3, 1).contiguous()                  # no-op for channels-last activations
        y = hipops.groupnorm_nhwc(xn, norm.weight, norm.bias, norm.eps, 1 if silu else 0)
        return y.permute(0, 3, 1, 2)
    if _native_expected(x):
        note_fallback("groupnorm", f"C={x.shape[1]} groups={norm.num_groups} affine={norm.weight.dtype}")
    y = norm(x)
    return F.silu(y) if silu else y


def _rows_kernel_ok(x, *params):
    """token-matrix HIP kernels (LayerNorm, GEGLU) are forward-only: used when nothing on this path needs autograd
    (the diffusion nets in score distillation); otherwise the ATen ops run on the same device."""
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    return x.is_cuda and x.dtype in HALF and CONV_BACKEND == "mfma" and not needs_grad


_WT_CACHE = {}        # id(weight) -> ((data_ptr, version, dtype), w contiguous, w^T contiguous) of frozen layers differentiated through


def linear_fused(x, weight, bias, residual=None):
    """F.linear(x, weight, bias) (+ residual): frozen bf16 layers on the GPU run the hand-written MFMA GEMM with the adds in
    its epilogue (csrc/conv.hip, 1-tap instantiation); everything else is the ATen call."""
    N, K = weight.shape
    if (_rows_kernel_ok(x, weight) and weight.dtype == x.dtype and hipops.gemm_fused_ok(x.numel() // K, K, N)
            and (residual is None or residual.dtype == x.dtype)):
        r = residual.contiguous() if residual is not None else None
        return hipops.gemm_fused(x.contiguous(), weight.detach().contiguous(), bias.detach() if bias is not None else None, r)
    M = x.numel() // K
    if (CONV_BACKEND == "mfma" and x.is_cuda and x.dtype in HALF and weight.dtype == x.dtype and torch.is_grad_enabled() and x.requires_grad
            and not weight.requires_grad and (bias is None or not bias.requires_grad)
            and hipops.gemm_fused_ok(M, K, N) and hipops.gemm_fused_ok(M, N, K) and (residual is None or residual.dtype == x.dtype)):
        # differentiated activations through a FROZEN layer (the VAE encoder's 1 x 1 shortcuts and attention projections): forward and
        # data gradient on the hand-written GEMM (hipops._LinearFrozen); w^T prepared once per weight tensor
        cache = _WT_CACHE.get(id(weight))
        key = (weight.data_ptr(), weight._version, weight.dtype)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, weight.detach().contiguous(), weight.detach().t().contiguous())
            if len(_WT_CACHE) > 64:
                _WT_CACHE.clear()
            _WT_CACHE[id(weight)] = cache
        r = residual.contiguous() if residual is not None else None
        return hipops.linear_frozen_autograd(x.contiguous(), cache[1], cache[2], bias.detach() if bias is not None else None, r)
    if (CONV_BACKEND == "mfma" and x.is_cuda and x.dtype in HALF and weight.dtype == x.dtype and residual is None
            and hipops.linear_small_ok(M, K, N) and not (torch.is_grad_enabled() and (weight.requires_grad or (bias is not None and bias.requires_grad)))):
        # a few-channel layer (AutoencoderKL's quant_conv, 8 -> 8): its own small kernel, forward and data gradient
        wd = weight.detach().contiguous()
        bd = bias.detach() if bias is not None else None
        if torch.is_grad_enabled() and x.requires_grad:
            return hipops.linear_small_autograd(x.contiguous(), wd, wd.t().contiguous(), bd)
        return hipops.linear_small(x.contiguous(), wd, bd)
    if _native_expected(x):
        grad = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)
        note_fallback("linear", f"K={K} N={N} M={x.numel() // K}{' autograd' if grad else ''}")
    y = F.linear(x, weight, bias)
    return y + residual if residual is not None else y


def layer_norm(norm: nn.LayerNorm, x):
    C = x.shape[-1]
    if _rows_kernel_ok(x, norm.weight) and C % 8 == 0 and C <= 2048:
        return hipops.layernorm_rows(x.contiguous(), norm.weight, norm.bias, norm.eps)
    if _native_expected(x):
        note_fallback("layernorm", f"C={C}{' autograd' if torch.is_grad_enabled() and x.requires_grad else ''}")
    return norm(x)


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_ch, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_ch, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class PaddedContext:
    """encoder_hidden_states zero-padded to a multiple of 8 tokens (16 B rows for V^T) + true length.

    `bank` / `ids` (optional): the rows of the batch are `bank[ids]` with `bank` a SMALL set of distinct embeddings that
    does not change between steps (DreamMat: 4 view-dependent prompts x {text, negative} + the empty prompt = 9, drawn 3 x
    views times per step).  The nets are frozen, so the K / V^T projections of the bank are computed ONCE per cross-attention
    layer and a step only gathers them (Attention.forward): 46 projection GEMMs per step become 46 row gathers."""

    def __init__(self, ctx, bank=None, ids=None):
        B, S, C = ctx.shape
        self.len = S
        pad = (-S) % 8
        self.t = F.pad(ctx, (0, 0, 0, pad)) if pad else ctx
        self.bank, self.ids = None, None
        self.gathered = {}                     # id(Attention) -> (bank entry key, K rows, V^T rows) of this step (NetPrologue)
        if bank is not None and ids is not None:
            self.bank = F.pad(bank, (0, 0, 0, pad)) if pad else bank
            self.ids = ids
            self.bank_key = (bank.data_ptr(), bank._version, tuple(bank.shape), bank.dtype)


def _zero_tail(vt, kv_len):
    vt[:, :, kv_len:] = 0
    return vt


def project_vt(v_weight, v_bias, kv_src, kv_len):
    """V^T = (kv_src @ v_weight^T + bias)^T as [B, C, Skv_pad] bf16 with the padded tail zeroed (the layout the MFMA kernels take)."""
    C = v_weight.shape[0]
    if kv_src.shape[1] % 8:                                       # V^T rows must be 16 B multiples
        kv_src = F.pad(kv_src, (0, 0, 0, (-kv_src.shape[1]) % 8))
    Bk, Sk = kv_src.shape[0], kv_src.shape[1]
    if C >= 640 and Bk >= 16 and Sk * C >= (1 << 20) and hipops.gemm_fused_ok(Bk * Sk, kv_src.shape[2], C):
        # hipBLASLt / rocBLAS in this image fault (HIPBLAS_STATUS_INTERNAL_ERROR, then an illegal address) on the strided-
        # batched W[C,C] @ X[B,S,C]^T for 24 x 4096 x 640, 24 x 1024 x 1280, 24 x 4096 x 1280 ... (tools/blas_probe.py:
        # the 1024^2 configurations; every shape of the 512^2 step is fine): there V comes from the fused GEMM kernel
        # and is transposed by a copy
        vt = linear_fused(kv_src, v_weight, v_bias).transpose(1, 2).contiguous()
        return vt if kv_len == Sk else _zero_tail(vt, kv_len)
    if (VT_BY_FUSED_GEMM and v_bias is None and kv_len == Sk and kv_src.is_contiguous()
            and _rows_kernel_ok(kv_src, v_weight) and hipops.gemm_fused_ok(C, kv_src.shape[2], Bk * Sk)):
        # V^T straight from the hand-written GEMM with the operands swapped (round 4): Y[C, B S] = W_v . X_all^T -- the weight
        # matrix plays the activations ([M = C, K]) and the token rows of ALL batch items play the weights ([N = B S, K]), so
        # row c of Y is channel c of V for every token, i.e. batch item b's V^T is the column block [b S, (b + 1) S) with a
        # row pitch of B S: exactly the strided [B, C, S] view the attention kernels take.  One launch per layer, no hipBLASLt.
        y = hipops.gemm_fused(v_weight.detach().contiguous(), kv_src.view(Bk * Sk, kv_src.shape[2]), None, None)      # [C, B S]
        return y.view(C, Bk, Sk).permute(1, 0, 2)
    vt = torch.matmul(v_weight, kv_src.transpose(1, 2))           # [B, C, Skv_pad]: V^T for free
    if v_bias is not None:
        vt = vt + v_bias[None, :, None]
        if kv_len < vt.shape[2]:
            vt[:, :, kv_len:] = 0
    return vt


def set_attention_precision(net, precision):
    """guidance.attention_precision -> the Attention modules of ONE net (a captured hipGraph keeps what its net had at capture)."""
    if precision not in ("16bit", "fp8"):
        raise ValueError(f"attention precision must be '16bit' or 'fp8', got '{precision}'")
    for m in net.modules():
        if isinstance(m, Attention):
            m.attention_precision = precision


def attention_core(q, k, v_weight, v_bias, kv_src, heads, kv_len, precision=None):
    """softmax(QK^T/sqrt(d)) V with V = kv_src @ v_weight^T (+bias).  q [B,Sq,C], k [B,Skv,C].
    CUDA+bf16 -> MFMA kernels (inference: V produced directly transposed by the projection GEMM; under autograd:
    hipops.attention_train, whose backward recomputes the probabilities); otherwise plain fp32-style math."""
    B, Sq, C = q.shape
    D = C // heads
    needs_grad = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or kv_src.requires_grad
                                              or v_weight.requires_grad)
    if q.is_cuda and q.dtype in HALF and not needs_grad:     # inference: V arrives transposed from its projection
        vt = project_vt(v_weight, v_bias, kv_src, kv_len)
        if ((precision or ATTENTION_PRECISION) == "fp8" and kv_len == k.shape[1] and kv_len >= FP8_MIN_SEQ and Sq >= FP8_MIN_SEQ
                and hipops.attention_fp8_ok(q, k, heads)):
            return hipops.attention_fp8(q, k, vt, heads)
        return hipops.attention(q, k[:, :kv_len], vt, heads)
    v = F.linear(kv_src, v_weight, v_bias)
    if q.is_cuda and TRAIN_KERNELS and hipops.attention_train_ok(q, k, v, heads):     # differentiated (ControlNet training): MFMA fwd + bwd
        return hipops.attention_train(q, k[:, :kv_len], v[:, :kv_len], heads)
    if _native_expected(q):
        note_fallback("attention", f"heads={heads} D={D} Sq={Sq} Skv={kv_len} autograd")
    qh = q.view(B, Sq, heads, D).transpose(1, 2)
    kh = k[:, :kv_len].reshape(B, kv_len, heads, D).transpose(1, 2)
    vh = v[:, :kv_len].reshape(B, kv_len, heads, D).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-1, -2)) * (D ** -0.5)
    p = torch.softmax(s.float(), dim=-1).to(q.dtype)
    return torch.matmul(p, vh).transpose(1, 2).reshape(B, Sq, C)


class Attention(nn.Module):
    """diffusers `Attention` (to_q/to_k/to_v without bias, to_out.0 with bias)."""

    def __init__(self, query_dim, heads, cross_dim=None, bias=False):
        super().__init__()
        self.heads = heads
        cross_dim = cross_dim or query_dim
        self.to_q = nn.Linear(query_dim, query_dim, bias=bias)
        self.to_k = nn.Linear(cross_dim, query_dim, bias=bias)
        self.to_v = nn.Linear(cross_dim, query_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Identity()])

    def _qk_fused_weight(self, x):
        """[Wq; Wk] for the fused projection of a frozen self-attention layer, or None when the two run on their own"""
        wq, wk = self.to_q.weight, self.to_k.weight
        C, K = wq.shape
        if not (QK_FUSED and C % 128 == 0 and 2 * C % 256 == 0 and C >= 640 and wk.shape == wq.shape and self.to_q.bias is None
                and self.to_k.bias is None and _rows_kernel_ok(x, wq, wk, self.to_v.weight) and wq.dtype == x.dtype and wk.dtype == x.dtype
                and hipops.gemm_fused_ok(x.numel() // K, K, 2 * C)):
            return None
        key = (wq.data_ptr(), wq._version, wk.data_ptr(), wk._version, wq.dtype)
        if self.__dict__.get("_wqk_key") != key:
            self.__dict__["_wqk"] = torch.cat([wq.detach(), wk.detach()]).contiguous()
            self.__dict__["_wqk_key"] = key
        return self.__dict__["_wqk"]

    def forward(self, x, context=None, residual=None):
        """-> to_out(attention) (+ residual, added in the output projection's epilogue)."""
        if context is None:
            src, kv_len = x, x.shape[1]
        else:
            src, kv_len = context.t, context.len
        wqk = self._qk_fused_weight(x) if context is None else None
        if wqk is not None:
            # self-attention, frozen: q | k from ONE GEMM (x is read once; tools/gemm_fit.sh: 2 x 40 -> 62 us at C = 640 and
            # 2 x 33 -> 50 us at C = 1280 per layer; at C = 320 two single-tile N = 320 launches are the faster form, so not there).
            # q and k are row-strided views of its output -- the attention kernels take strides.
            C = self.to_q.weight.shape[0]
            qk = hipops.gemm_fused(x.contiguous(), wqk, None, None)
            return linear_fused(attention_core(qk[..., :C], qk[..., C:], self.to_v.weight, self.to_v.bias, src, self.heads, kv_len,
                                               getattr(self, "attention_precision", None)),
                                self.to_out[0].weight, self.to_out[0].bias, residual)
        q = linear_fused(x, self.to_q.weight, self.to_q.bias)
        frozen = not (torch.is_grad_enabled() and (self.to_k.weight.requires_grad or self.to_v.weight.requires_grad))
        if (context is not None and context.bank is not None and frozen and x.is_cuda and x.dtype in HALF
                and not (torch.is_grad_enabled() and x.requires_grad)):
            # frozen projections of a fixed bank of prompt embeddings: computed once, gathered per step
            key = (context.bank_key, self.to_k.weight.data_ptr(), self.to_k.weight._version, self.to_v.weight.data_ptr(),
                   self.to_v.weight._version)
            # keyed by (bank, weights): EVERY entry stays alive, because a captured hipGraph (guidance.hip_graph) bakes the raw
            # addresses of the entry it was captured with -- replacing a single slot when a second bank shows up ("vd" vs
            # "plain", a re-created prompt processor) would let an older graph replay against freed memory.
            # A bank is ~9 x 77 x C values per layer; reloading the weights changes the key and drops the stale entries.
            cache = self.__dict__.setdefault("_kv_banks", {})
            if key not in cache:
                for old in [k for k in cache if k[1:] != key[1:]]:      # other weights: those entries can never be hit again
                    del cache[old]
                with torch.no_grad():
                    cache[key] = (linear_fused(context.bank, self.to_k.weight, self.to_k.bias),
                                  project_vt(self.to_v.weight, self.to_v.bias, context.bank, kv_len))
            pre = context.gathered.pop(id(self), None)
            if pre is not None and pre[0] == key:
                # this step's rows of the bank, gathered for all layers of one width at once (NetPrologue.gather_kv)
                o = hipops.attention(q, pre[1][:, :kv_len], pre[2], self.heads)
            else:
                kb, vtb = cache[key]
                o = hipops.attention(q, kb.index_select(0, context.ids)[:, :kv_len], vtb.index_select(0, context.ids), self.heads)
        else:
            k = linear_fused(src, self.to_k.weight, self.to_k.bias)
            o = attention_core(q, k, self.to_v.weight, self.to_v.bias, src, self.heads, kv_len, getattr(self, "attention_precision", None))
        return linear_fused(o, self.to_out[0].weight, self.to_out[0].bias, residual)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def _interleaved(self):
        w = self.proj.weight
        key = (w.data_ptr(), w._version, w.dtype)
        if getattr(self, "_il_key", None) != key:
            self._w_il = hipops.geglu_interleave(w.detach())
            self._b_il = hipops.geglu_interleave(self.proj.bias.detach()) if self.proj.bias is not None else None
            self._il_key = key
        return self._w_il, self._b_il

    def forward(self, x):
        w = self.proj.weight
        N, K = w.shape
        if (_rows_kernel_ok(x, w) and w.dtype == x.dtype and (N // 2) % 32 == 0
                and hipops.gemm_fused_ok(x.numel() // K, K, N, geglu=True)):
            w_il, b_il = self._interleaved()                       # GEMM + GEGLU in one kernel: the 2x wide tensor never exists
            return hipops.gemm_fused(x.contiguous(), w_il, b_il, None, geglu=True)
        if _native_expected(x):
            note_fallback("geglu", f"K={K} N={N} M={x.numel() // K}{' autograd' if torch.is_grad_enabled() and x.requires_grad else ''}")
        h = self.proj(x)
        if _rows_kernel_ok(h) and h.shape[-1] % 16 == 0:
            return hipops.geglu_rows(h.contiguous())
        x, gate = h.chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x, residual=None):
        return linear_fused(self.net[0](x), self.net[2].weight, self.net[2].bias, residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = self.attn1(layer_norm(self.norm1, x), None, x)         # the three residual adds ride in GEMM epilogues
        x = self.attn2(layer_norm(self.norm2, x), context, x)
        return self.ff(layer_norm(self.norm3, x), x)


class Transformer2DModel(nn.Module):
    def __init__(self, ch, heads, cross_dim, use_linear_projection):
        super().__init__()
        self.use_linear = use_linear_projection
        self.norm = nn.GroupNorm(32, ch, eps=1e-6)
        self.proj_in = nn.Linear(ch, ch) if use_linear_projection else Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, cross_dim)])
        self.proj_out = nn.Linear(ch, ch) if use_linear_projection else Conv2d(ch, ch, 1)

    def forward(self, x, context):
        B, C, H, W = x.shape
        res = x
        h = group_norm_act(self.norm, x, False)
        if self.use_linear:
            h = linear_fused(h.permute(0, 2, 3, 1).reshape(B, H * W, C), self.proj_in.weight, self.proj_in.bias)
        else:
            h = self.proj_in(h).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        if self.use_linear:
            r = res.permute(0, 2, 3, 1).reshape(B, H * W, C)      # a view for channels-last activations
            h = linear_fused(h, self.proj_out.weight, self.proj_out.bias, r)
            return h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = self.proj_out(h.reshape(B, H, W, C).permute(0, 3, 1, 2))
        return h + res


class NetPrologue:
    """Work that every block of a frozen UNet / ControlNet would otherwise launch for itself, done once per forward for the whole
    net (diffusers computes both per block: ResnetBlock2D.forward `self.time_emb_proj(self.nonlinearity(temb))`, Attention's
    to_k / to_v on encoder_hidden_states; reached from dreammat_guidance.py:205-292).  At one view per rank a step is ~1400
    launches of 5-20 us: 34 SiLU + 34 M = 3 B GEMMs and 46 row gathers are 114 of them.
      project_temb: the time-embedding projections of all ResnetBlock2D of one output width in ONE batched product
      gather_kv:    this step's rows of the cross-attention K / V^T banks (Attention.forward) of all layers of one width in ONE
                    index_select each
    Weights are read through per-net caches keyed by (data_ptr, _version) of every tensor involved, like Conv2d._prepared."""

    def __init__(self, net: nn.Module):
        self.net = net
        self.resnets = [m for m in net.modules() if isinstance(m, ResnetBlock2D) and m.time_emb_proj is not None]
        self.cross = [m for n, m in net.named_modules() if isinstance(m, Attention) and n.endswith("attn2")]
        self._temb_key, self._temb_groups = None, []
        self._kv = {}                          # key -> groups; every entry stays alive (captured hipGraphs bake addresses)

    @staticmethod
    def usable(t):
        return t.is_cuda and t.dtype in HALF and CONV_BACKEND == "mfma"

    def project_temb(self, temb):
        for m in self.resnets:                  # a projection left behind by a forward that raised midway must never be consumed
            m.__dict__.pop("_tproj", None)
        if not self.resnets or not self.usable(temb):
            return
        if torch.is_grad_enabled() and (temb.requires_grad or any(m.time_emb_proj.weight.requires_grad for m in self.resnets)):
            return                              # (ControlNet training: the layers run themselves, under autograd)
        key = tuple((m.time_emb_proj.weight.data_ptr(), m.time_emb_proj.weight._version, m.time_emb_proj.bias.data_ptr(),
                     m.time_emb_proj.bias._version) for m in self.resnets) + (temb.dtype,)
        if key != self._temb_key:
            by_width = {}
            for m in self.resnets:
                by_width.setdefault(m.time_emb_proj.out_features, []).append(m)
            with torch.no_grad():
                self._temb_groups = [(ms, torch.stack([m.time_emb_proj.weight.detach().t() for m in ms]).contiguous(),
                                      torch.stack([m.time_emb_proj.bias.detach() for m in ms]).unsqueeze(1).contiguous())
                                     for ms in by_width.values()]
            self._temb_key = key
        act = F.silu(temb)
        for ms, w, b in self._temb_groups:
            out = torch.baddbmm(b, act.unsqueeze(0).expand(len(ms), -1, -1), w)       # [G, B, C]
            for g, m in enumerate(ms):
                m._tproj = out[g]

    def gather_kv(self, context):
        if (not self.cross or context is None or context.bank is None or not self.usable(context.bank)
                or torch.is_grad_enabled() and any(m.to_k.weight.requires_grad or m.to_v.weight.requires_grad for m in self.cross)):
            return
        keys = [(context.bank_key, m.to_k.weight.data_ptr(), m.to_k.weight._version, m.to_v.weight.data_ptr(),
                 m.to_v.weight._version) for m in self.cross]
        if any(k not in m.__dict__.get("_kv_banks", {}) for k, m in zip(keys, self.cross)):
            return                              # first forward with this bank: the layers build their entries, the next one groups them
        gkey = tuple(keys)
        if gkey not in self._kv:
            for old in [k for k in self._kv if tuple(e[1:] for e in k) != tuple(e[1:] for e in gkey)]:   # other weights
                del self._kv[old]
            by_width = {}
            for k, m in zip(keys, self.cross):
                by_width.setdefault(m.to_q.out_features, []).append((k, m))
            with torch.no_grad():
                self._kv[gkey] = [([k for k, _ in km], [m for _, m in km],
                                   torch.stack([m._kv_banks[k][0] for k, m in km]).contiguous(),
                                   torch.stack([m._kv_banks[k][1] for k, m in km]).contiguous()) for km in by_width.values()]
        for ks, ms, kb, vtb in self._kv[gkey]:
            kr, vr = kb.index_select(1, context.ids), vtb.index_select(1, context.ids)    # [L, B, S, C], [L, B, C, S]
            for l, (k, m) in enumerate(zip(ks, ms)):
                context.gathered[id(m)] = (k, kr[l], vr[l])


class ResnetBlock2D(nn.Module):
    def __init__(self, in_ch, out_ch, temb_ch=1280, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, in_ch, eps=eps)
        self.conv1 = Conv2d(in_ch, out_ch, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, out_ch) if temb_ch else None
        self.norm2 = nn.GroupNorm(32, out_ch, eps=eps)
        self.conv2 = Conv2d(out_ch, out_ch, 3, padding=1)
        self.conv_shortcut = Conv2d(in_ch, out_ch, 1) if in_ch != out_ch else None

    def forward(self, x, temb=None):
        needs_grad = torch.is_grad_enabled() and x.requires_grad
        if self.conv1.gn_fold_ok(self.norm1, x) and (not needs_grad or (temb is None and self.time_emb_proj is None)):
            # round 6: both GroupNorm apply passes ride in the convolutions they feed (no normalised tensor is written or read);
            # under autograd (VAE encoder) x reaches norm1 and the skip through ONE node, as in the branch below
            tproj = self.__dict__.pop("_tproj", None)
            if tproj is None and self.time_emb_proj is not None:
                tproj = self.time_emb_proj(F.silu(temb))
            if needs_grad:
                h, x = self.conv1.forward_gn_fold(self.norm1, x, True, with_skip=True)
            else:
                h = self.conv1.forward_gn_fold(self.norm1, x, True, rowbias=tproj)
            res = self.conv_shortcut(x) if self.conv_shortcut is not None else x
            if self.conv2.gn_fold_ok(self.norm2, h):
                return self.conv2.forward_gn_fold(self.norm2, h, True, residual=res)
            h = group_norm_act(self.norm2, h, True)
            return self.conv2.forward_fused(h, residual=res) if self.conv2.fused_ok(h) else self.conv2.forward_residual(h, res)
        if torch.is_grad_enabled() and x.requires_grad and _gn_kernel_ok(self.norm1, x) and not self.norm1.weight.requires_grad:
            # differentiated (VAE encoder) and x feeds norm1 AND the skip (directly or through conv_shortcut): both gradients meet
            # in the GroupNorm backward kernel
            h, x = hipops.groupnorm_nhwc_skip(x.permute(0, 2, 3, 1).contiguous(), self.norm1.weight, self.norm1.bias, self.norm1.eps, 1)
            h, x = h.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2)
        else:
            h = group_norm_act(self.norm1, x, True)
        tproj = self.__dict__.pop("_tproj", None)                  # left here by NetPrologue.project_temb for this forward
        if tproj is None and self.time_emb_proj is not None:
            tproj = self.time_emb_proj(F.silu(temb))
        if self.conv1.fused_ok(h) and self.conv2.fused_ok(h):
            # inference path (diffusion nets in SDS): both adds ride in the conv epilogues
            h = self.conv1.forward_fused(h, rowbias=tproj)
            h = group_norm_act(self.norm2, h, True)
            if self.conv_shortcut is not None:
                x = self.conv_shortcut(x)
            return self.conv2.forward_fused(h, residual=x)
        h = self.conv1(h)
        if tproj is not None:
            h = h + tproj[:, :, None, None]
        h = group_norm_act(self.norm2, h, True)
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return self.conv2.forward_residual(h, x)


class Downsample2D(nn.Module):
    def __init__(self, ch, asymmetric_pad=False):
        super().__init__()
        self.asym = asymmetric_pad
        self.conv = Conv2d(ch, ch, 3, stride=2, padding=0 if asymmetric_pad else 1)

    def forward(self, x):
        if self.asym:
            Cout, Cin = self.conv.weight.shape[:2]
            if self.conv.mfma_ok(x) and Cin % 64 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
                return self.conv.forward_strided_asym(x)
            x = F.pad(x, (0, 1, 0, 1))
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        conv = self.conv
        Cout, Cin = conv.weight.shape[:2]
        mode = os.environ.get("DREAMMAT_UPSAMPLE", "auto")          # auto | subpixel | materialize (tests, tools/upsample_time.py)
        # measured (24 images): 640 @32 -> 64: 463-542 us against 763-801 for upsample + 3x3; 1280 @16 -> 32 equal; 1280 @8 -> 16 slower
        # (1944 rows against N = K = 5120 leave the 256 x 256 tiles a fraction of the chip) -- so only from 16 k source pixels on
        big = x.shape[0] * x.shape[2] * x.shape[3] >= 16384
        if (conv.mfma_ok(x) and Cin % 64 == 0 and Cout % 64 == 0 and not (torch.is_grad_enabled() and x.requires_grad)
                and (mode == "subpixel" or (mode == "auto" and big))):
            # no 4x larger tensor: the 3x3 taps that fall on one source pixel are summed into a 2 x 2 window at the source
            # resolution, the four output parities are four channel blocks (hipops.subpixel_upsample_weights)
            w = conv.weight
            key = (w.data_ptr(), w._version, w.dtype)
            if getattr(self, "_sub_key", None) != key:
                with torch.no_grad():
                    self._w4, self._b4 = hipops.subpixel_upsample_weights(w.detach(), conv.bias.detach() if conv.bias is not None else None)
                self._sub_key = key
            return hipops.conv3x3_upsampled_nhwc(x.permute(0, 2, 3, 1).contiguous(), self._w4, self._b4).permute(0, 3, 1, 2)
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
