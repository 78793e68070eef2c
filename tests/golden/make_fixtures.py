"""Regenerates the committed fixtures.  Runs ONLY in the development container (it
reads the reference's scene *data* files under /root/reference/scenes); the GPU box
gets the resulting files with the repository snapshot.

  killeroo_simple.hprt   baked scene (post-parse, world space) of scenes/killeroo-simple
                         with $acc="bvh": camera/film/sampler/integrator parameters,
                         materials, the sphere area light and the two Loop-subdivided
                         killeroo meshes plus the two quads, in primitive-creation order.
  killeroo_simple_8spp_srgb8.npz  the reference's own checked-in render of that scene
                         (scenes/killeroo-simple.png, 8 spp, 8-bit sRGB) as a uint8 array —
                         a data file the reference ships as its regression image
                         (scripts/render.sh:4).
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/scenes"


def main():
    hprt = importlib.import_module("thesis-pbrt-v3_amd")
    m = hprt.Model.parse(os.path.join(REF, "killeroo-simple"),
                         {"$acc": '"bvh"', "/Programming/Thesis/pbrt-v3/scenes/": REF + "/"})
    print("killeroo-simple:", m.counts(), m.warnings())
    m.save(os.path.join(HERE, "killeroo_simple.hprt"))
    from PIL import Image
    png = np.asarray(Image.open(os.path.join(REF, "killeroo-simple.png")).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "killeroo_simple_8spp_srgb8.npz"), srgb8=png)


if __name__ == "__main__":
    main()
