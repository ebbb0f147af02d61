"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain C for integer/per-pixel work, fp32 torch/numpy for the rest) of the
reference's SDS material-fitting step (zzzyuqing/DreamMat, threestudio_dreammat/...).  Each
function cites the reference file:line it follows.

PARITY STATUS
  * in-tree pure-python arithmetic (camera math, lin2srgb, material activation + smoothness
    regulariser, split-sum composition, the Monte-Carlo ray-traced shading branch with its sampling /
    pdf / BRDF helpers incl. autograd gradients, ControlNet-normal/depth encodings, SDS gradient, the
    C() schedule evaluator) is PINNED: tests/golden/*.npz were produced by executing the
    reference's own function bodies (tests/golden/make_golden.py AST-extracts them from
    /root/reference) and the oracle is checked against them.
  * arithmetic living in un-vendored dependencies (nvdiffrast rasterize/interpolate/antialias/
    texture, envlight, tiny-cuda-nn HashGrid, the `raytracing` BVH tracer, diffusers
    UNet/ControlNet/VAE/DDIM) is restated
    from their published behaviour: PARITY UNPINNED against the real packages (none of them is
    installable here: CUDA-only / no network; the reference ships no tests or golden vectors).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (dreammat_amd/) never does.
"""
