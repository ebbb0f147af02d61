"""Writers of the train/val/test/export hooks: images (threestudio/utils/saving.py:301-334: `(x*255).astype(uint8)`
through PIL; grids are rows of equally sized images) and Wavefront OBJ + MTL + texture maps (saving.py:456-655)."""
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


def save_gif(img_dir, n_frames=120, fps=30, name="eval.gif"):
    """saving.py:401-408: frames `<img_dir>/<i>.png`, i = 0..n_frames-1 (the reference hard-codes 120 = its test view count),
    written as `<img_dir>/eval.gif` at `fps` (imageio there, PIL here; frames that are missing end the sequence early)."""
    frames = []
    for i in range(int(n_frames)):
        f = os.path.join(img_dir, f"{i}.png")
        if not os.path.exists(f):
            break
        frames.append(Image.open(f).convert("RGB"))
    if not frames:
        raise FileNotFoundError(f"no frames 0.png.. under {img_dir}")
    out = os.path.join(img_dir, name)
    frames[0].save(out, save_all=True, append_images=frames[1:], duration=max(1, int(round(1000.0 / fps))), loop=0)
    return out


def _np(x):
    return x.detach().float().cpu().numpy() if torch.is_tensor(x) else x


def save_mtl(path, matname, map_Kd=None, map_Ks=None, map_Bump=None, map_Pm=None, map_Pr=None, map_format="jpg",
             Ka=(0.0, 0.0, 0.0), Kd=(1.0, 1.0, 1.0), Ks=(0.0, 0.0, 0.0)):
    """saving.py:561-655: texture files sit next to the .mtl as texture_{kd,ks,nrm,metallic,roughness}.<fmt>."""
    out = [path]
    d = os.path.dirname(path)
    lines = [f"newmtl {matname}", f"Ka {Ka[0]} {Ka[1]} {Ka[2]}"]

    def tex(img, stem, key, rgb):
        fn = f"texture_{stem}.{map_format}"
        a = _u8(img)
        if not rgb:
            a = a[..., 0]
        Image.fromarray(a).save(os.path.join(d, fn))
        lines.append(f"{key} {fn}")
        out.append(os.path.join(d, fn))
    if map_Kd is not None:
        tex(map_Kd, "kd", "map_Kd", True)
    else:
        lines.append(f"Kd {Kd[0]} {Kd[1]} {Kd[2]}")
    if map_Ks is not None:
        tex(map_Ks, "ks", "map_Ks", True)
    else:
        lines.append(f"Ks {Ks[0]} {Ks[1]} {Ks[2]}")
    if map_Bump is not None:
        tex(map_Bump, "nrm", "map_Bump", True)
    if map_Pm is not None:
        tex(map_Pm, "metallic", "map_Pm", False)
    if map_Pr is not None:
        tex(map_Pr, "roughness", "map_Pr", False)
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    return out


def save_obj(path, mesh, save_mat=False, save_normal=False, save_uv=False, save_vertex_color=False, map_Kd=None,
             map_Ks=None, map_Bump=None, map_Pm=None, map_Pr=None, map_format="jpg"):
    """saving.py:456-559: `v x y z [r g b]`, `vn`, `vt u 1-v`, `f v/vt/vn` (1-based); returns the written paths."""
    if not path.endswith(".obj"):
        path += ".obj"
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    paths = []
    v, f = _np(mesh.v_pos), _np(mesh.t_pos_idx).astype(np.int64)
    vn = _np(mesh.v_nrm) if save_normal else None
    vt, ft = (_np(mesh.v_tex), _np(mesh.t_tex_idx).astype(np.int64)) if save_uv else (None, None)
    rgb = _np(mesh.v_rgb) if save_vertex_color else None
    lines = []
    if save_mat:
        mtl = path[:-4] + ".mtl"
        paths += save_mtl(mtl, "default", map_Kd, map_Ks, map_Bump, map_Pm, map_Pr, map_format)
        lines += [f"mtllib {os.path.basename(mtl)}", "g object", "usemtl default"]
    for i in range(len(v)):
        s = f"v {v[i][0]} {v[i][1]} {v[i][2]}"
        if rgb is not None:
            s += f" {rgb[i][0]} {rgb[i][1]} {rgb[i][2]}"
        lines.append(s)
    if vn is not None:
        lines += [f"vn {n[0]} {n[1]} {n[2]}" for n in vn]
    if vt is not None:
        lines += [f"vt {t[0]} {1.0 - t[1]}" for t in vt]
    for i in range(len(f)):
        s = "f"
        for j in range(3):
            s += f" {f[i][j] + 1}/"
            if vt is not None:
                s += f"{ft[i][j] + 1}"
            s += "/"
            if vn is not None:
                s += f"{f[i][j] + 1}"
        lines.append(s)
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    paths.append(path)
    return paths
