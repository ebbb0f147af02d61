"""Image writers of the train/val/test hooks (threestudio/utils/saving.py:301-334: `(x*255).astype(uint8)`
through PIL; grids are rows of equally sized images)."""
import os

import numpy as np
import torch
from PIL import Image


def _u8(img):
    a = img.detach().float().clamp(0, 1).cpu().numpy()
    return (a * 255.0).astype(np.uint8)


def save_image_grid(path, images):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    row = np.concatenate([_u8(i if i.shape[-1] == 3 else i.expand(-1, -1, 3)) for i in images], axis=1)
    Image.fromarray(row).save(path)
    return path


def save_rgba(path, rgb, alpha):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(np.concatenate([_u8(rgb), _u8(alpha)], axis=-1), "RGBA").save(path)
    return path
