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
  dodecahedron.hprt, killeroo.hprt + *_8spp_srgb8.npz
                         two more of the reference's regression pairs (scenes/dodecahedron[.png]: two plastic
                         dodecahedra, distant light; scenes/killeroo[.png]: one plastic killeroo, distant light —
                         killeroo-test1..4.png are the same image, byte for byte).  Template tokens replaced as
                         scripts/render_simple.sh does.
  simple_instanced.hprt  baked scene of scenes/simple (eight spheres in one object definition,
                         one ObjectInstance, distant light) — the reference's object-instancing
                         scene — with the camera and light of the version its checked-in render
                         shows: eye (-5,0,0) and light from (-1,0,0), the values the previous
                         revision of that scene carries (scenes/old/simple-bvh-all.pbrt:1,22).
  simple_8spp_srgb8.npz  scenes/simple.png (8 spp, 8-bit sRGB) as a uint8 array.  With the two
                         values above the oracle reproduces it bit for bit; with the current
                         scenes/simple text (eye (-5,-5,0)) it shows a different view.
  living_room.hprt       the GEOMETRY of scenes/livingroom (65 PLY meshes under scenes/living-room/models, 143,163
                         triangles with normals and uv; "The Grey & White Room" by Wig42, CC BY 3.0, converted by Benedikt
                         Bitterli — scenes/living-room/LICENSE.txt) as the reference's only asset-backed interior: a
                         real stand-in for BASELINE.json's conference-room configuration.  Everything outside the hot
                         path's scope is replaced here, in the text, before parsing: the environment light (its map is
                         not in the repository) by a point light under the ceiling, sobol by halton, the triangle filter
                         by the box filter, maxdepth 65 by 5, the two wood textures whose files are not in the
                         repository (.MISSING_LARGE_BLOBS) by a constant.  The two image files that ARE there are kept
                         (round 3): picture8.tga on the painting and leaf.tga on the leaves' Kd and — as the scene binds it,
                         scenes/livingroom:30 — on their uber OPACITY; they travel in the container as their 8-bit texels
                         (version 6, 3.6 MB) and the product rebuilds the MIPMaps at load.
                         Its matte (incl. the OrenNayar plant pot), substrate, metal, mirror, glass and uber materials are
                         kept: the front-end reports no substitution for this text.
                         No reference render exists for it (the
                         checked-in TungstenRender.png is another renderer's): it pins nothing, it is a workload.
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
    sub = {"$acc": '"bvh"', "$accnr": "0", "$splitalpha": "0", "$alphatype": "0", "$axisselectiontype": "0", "$axisselectionamount": "0",
           "/Programming/Thesis/pbrt-v3/scenes/": REF + "/", "../../../../scenes/": REF + "/"}
    for name in ("dodecahedron", "killeroo"):
        m = hprt.Model.parse(os.path.join(REF, name), sub)
        print(name + ":", m.counts(), m.warnings())
        m.save(os.path.join(HERE, name + ".hprt"))
        png = np.asarray(Image.open(os.path.join(REF, name + ".png")).convert("RGB"))
        np.savez_compressed(os.path.join(HERE, name + "_8spp_srgb8.npz"), srgb8=png)
    import tempfile
    text = open(os.path.join(REF, "simple")).read()
    text = text.replace("LookAt -5 -5 0", "LookAt -5 0 0").replace('"point from" [-1 -1 0]', '"point from" [-1 0 0]')
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "simple.pbrt")
        open(p, "w").write(text)
        m = hprt.Model.parse(p, {"$acc": '"bvh"'})
    print("simple:", m.counts(), m.warnings())
    m.save(os.path.join(HERE, "simple_instanced.hprt"))
    png = np.asarray(Image.open(os.path.join(REF, "simple.png")).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "simple_8spp_srgb8.npz"), srgb8=png)
    living_room(hprt)


def living_room(hprt):
    import re, tempfile
    text = open(os.path.join(REF, "livingroom")).read()
    text = text.replace("/Programming/Thesis/pbrt-v3/scenes/", REF + "/")
    text = re.sub(r'Accelerator \$acc[^\n]*', 'Accelerator "bvh"', text)
    text = re.sub(r'Integrator "path"[^\n]*', 'Integrator "path" "integer maxdepth" [ 5 ]', text)
    text = re.sub(r'Sampler "sobol"', 'Sampler "halton"', text)
    text = re.sub(r'PixelFilter "triangle"[^\n]*', 'PixelFilter "box" "float xwidth" [ 0.5 ] "float ywidth" [ 0.5 ]', text)
    # Texture01 / 02 (wood.tga, wood5.tga) are stripped from the reference repository (.MISSING_LARGE_BLOBS): a constant stands in.
    # Texture03 (picture8.tga on "Painting") and Texture04 / 05 (leaf.tga on the leaves' Kd AND opacity) are there and are kept.
    text = re.sub(r'[ \t]*Texture "Texture0[12]"[^\n]*\n', '', text)
    text = re.sub(r'"texture Kd" \[ "Texture0[12]" \]', '"rgb Kd" [ 0.45 0.33 0.22 ]', text)
    text, n = re.subn(r'LightSource "infinite"[^\n]*', 'LightSource "point" "point from" [ 2.3 2.6 -1.5 ] "color I" [ 9 9 8.5 ]', text)
    assert n == 1
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "livingroom.pbrt")
        open(p, "w").write(text)
        m = hprt.Model.parse(p, {})
    print("living room:", m.counts(), len(m.warnings()), "warnings:", sorted(set(w[:60] for w in m.warnings())))
    assert m.counts()["textures"] == 2      # picture8.tga and leaf.tga (the texture cache gives Kd and opacity of the leaves ONE MIPMap)
    # compact: the two images travel as their 8-bit texels (3.6 MB), not as float pyramids (38 MB); rebuilt at load
    m.save(os.path.join(HERE, "living_room.hprt"), compact=True)


if __name__ == "__main__":
    main()
