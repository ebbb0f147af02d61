"""Copies the reference's own in-tree DATA assets used by BASELINE configs[1] (cfg2) into tests/golden/assets/ so the
GPU box (which has no /root/reference) can feed them to the HIP path:
    load/shapes/objs/apple.obj              (run_examples.sh:2, shape_init_params 0.7)
    load/lights/mud_road_puresky_1k.hdr     (CC0, polyhaven; DreamMatMaterial.Config.environment_texture default)
    load/lights/bsdf_256_256.bin            (the split-sum FG LUT, dreammat_material.py:399-404)
These are data fixtures (a mesh, an HDR probe, a float table), not source code.  Run from the repo root:
    python tests/golden/make_assets.py
"""
import hashlib
import json
import os
import shutil

REF = "/root/reference/threestudio_dreammat/load"
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
FILES = {"apple.obj": "shapes/objs/apple.obj", "mud_road_puresky_1k.hdr": "lights/mud_road_puresky_1k.hdr",
         "bsdf_256_256.bin": "lights/bsdf_256_256.bin", "LICENSE.txt": "lights/LICENSE.txt"}

if __name__ == "__main__":
    os.makedirs(HERE, exist_ok=True)
    sums = {}
    for name, rel in FILES.items():
        shutil.copyfile(os.path.join(REF, rel), os.path.join(HERE, name))
        os.chmod(os.path.join(HERE, name), 0o644)
        sums[name] = hashlib.sha256(open(os.path.join(HERE, name), "rb").read()).hexdigest()
    json.dump(sums, open(os.path.join(HERE, "SHA256.json"), "w"), indent=1)
    print(sums)
