"""Weight loading for the SD components from a local diffusers-layout directory
(<root>/{unet,vae}/diffusion_pytorch_model.safetensors, <controlnet_path>/diffusion_pytorch_model.safetensors),
i.e. what `from_pretrained(..., cache_dir=...)` resolves to in
threestudio/models/guidance/dreammat_guidance.py:110-129.  There is no network here: when a file is
absent the component keeps its seeded random initialisation and says so (synthetic benchmark mode)."""
import glob
import os

import torch

_OLD_VAE_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def _find(root, sub):
    if root is None:
        return None
    cands = [os.path.join(root, sub, "diffusion_pytorch_model.safetensors"),
             os.path.join(root, sub, "diffusion_pytorch_model.bin"),
             os.path.join(root, "diffusion_pytorch_model.safetensors"),
             os.path.join(root, "diffusion_pytorch_model.bin")]
    # huggingface cache layout: <cache>/models--org--name/snapshots/<hash>/<sub>/...
    cands += glob.glob(os.path.join(root, "**", sub, "diffusion_pytorch_model.safetensors"), recursive=True)
    for c in cands:
        if os.path.isfile(c):
            return c
    return None


def _read(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu")


def load_component(module, root, sub, strict=True, verbose=True):
    """Returns True if real weights were loaded, False if the random init was kept."""
    path = _find(root, sub)
    if path is None:
        if verbose:
            print(f"[dreammat_amd] no weights for '{sub}' under {root!r}: keeping seeded random init (synthetic mode)")
        return False
    sd = _read(path)
    if sub == "vae":
        fixed = {}
        for k, v in sd.items():
            if not (k.startswith("encoder.") or k.startswith("quant_conv.")):
                continue                                # decoder / post_quant_conv are not on the path
            for old, new in _OLD_VAE_ATTN.items():
                k = k.replace(f"attentions.0.{old}.", f"attentions.0.{new}.")
            if "attentions.0.to_" in k and k.endswith("weight") and v.dim() == 4:
                v = v[:, :, 0, 0]                       # very old checkpoints stored 1x1 convs
            fixed[k] = v
        sd = fixed
    module.load_state_dict(sd, strict=strict)
    return True
