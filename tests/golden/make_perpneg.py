"""Golden vectors for Perp-Neg prompting: the reference's own bodies of PromptProcessorOutput.get_text_embeddings_perp_neg
(prompt_processors/base.py:87-184), shifted_expotional_decay / perpendicular_component (utils/ops.py:423-441) executed on
seeded inputs (AST-extracted like make_golden.py; nothing of the reference is imported or copied).  Run from the repo root in
the build container:  python tests/golden/make_perpneg.py  -> tests/golden/perpneg.npz"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, extract, ns  # noqa: E402

if __name__ == "__main__":
    ops = extract(f"{REF}/utils/ops.py", ["shifted_expotional_decay", "perpendicular_component"])
    base_top = extract(f"{REF}/models/prompt_processors/base.py", ["shift_azimuth_deg"])
    meth = extract(f"{REF}/models/prompt_processors/base.py", ["get_text_embeddings_perp_neg"], cls="PromptProcessorOutput")
    env = ns()
    for src in list(ops.values()) + list(base_top.values()) + list(meth.values()):
        exec(src, env)
    g = torch.Generator().manual_seed(7)
    D = 16
    vd = torch.randn(4, 77, D, generator=g)
    uvd = torch.randn(4, 77, D, generator=g)
    null = torch.randn(1, 77, D, generator=g)
    ele = torch.tensor([10.0, 70.0, -5.0, 30.0, 0.0, 20.0])
    azi = torch.tensor([20.0, 100.0, -130.0, 200.0, 89.0, -91.0])
    dis = torch.full((6,), 3.5)
    shift = env["shift_azimuth_deg"]
    dirs = [types.SimpleNamespace(name="side", condition=lambda e, a, d: torch.ones_like(e, dtype=torch.bool)),
            types.SimpleNamespace(name="front", condition=lambda e, a, d: (shift(a) > -45) & (shift(a) < 45)),
            types.SimpleNamespace(name="back", condition=lambda e, a, d: (shift(a) > 135) | (shift(a) < -135)),
            types.SimpleNamespace(name="overhead", condition=lambda e, a, d: e > 60)]
    self = types.SimpleNamespace(directions=dirs, direction2idx={"side": 0, "front": 1, "back": 2, "overhead": 3},
                                 text_embeddings_vd=vd, uncond_text_embeddings_vd=uvd, null_text_embeddings=null,
                                 perp_neg_f_sb=(1, 0.5, -0.606), perp_neg_f_fsb=(1, 0.5, +0.967), perp_neg_f_fs=(4, 0.5, -2.426),
                                 perp_neg_f_sf=(4, 0.5, -2.426))
    emb, w = env["get_text_embeddings_perp_neg"](self, ele, azi, dis, True, True)
    x = torch.randn(3, 4, 8, 8, generator=g)
    y = torch.randn(3, 4, 8, 8, generator=g)
    perp = env["perpendicular_component"](x, y)
    np.savez(os.path.join(HERE, "perpneg.npz"), vd=vd, uvd=uvd, null=null, ele=ele, azi=azi, dis=dis, emb=emb, w=w, x=x, y=y, perp=perp)
    print(emb.shape, w)
