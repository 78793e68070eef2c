// hprt_render — a host program in C++ over nothing but include/hprt.h: parse (or load) a scene, build the reference's BVH,
// render it on one or several GPUs of this node and write the image.  This is the shape of the adapter a pbrt-side
// SamplerIntegrator::Render replacement has: one HprtScene per GPU, tile t of the 16x16 grid on GPU
// t mod N (core/integrator.cpp:237-244), hprt_film_gather_local as Film::MergeFilmTile, hprt_film_resolve +
// hprt_write_pfm as Film::WriteImage.  No Python, no torch: the library allocates its own device memory.
//
//   g++ -O2 -std=c++17 -pthread -Iinclude examples/hprt_render.cpp -o hprt_render -Lthesis-pbrt-v3_amd/lib -lhprt -Wl,-rpath,$PWD/thesis-pbrt-v3_amd/lib
//   ./hprt_render scene.pbrt|scene.hprt out.pfm [--spp N] [--gpus N] [--crop x0 x1 y0 y1]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "hprt.h"

#define TRY(call)                                                                                    \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != HPRT_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, hprt_last_error()); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s scene.pbrt|scene.hprt out.pfm [--spp N] [--gpus N] [--crop x0 x1 y0 y1]\n", argv[0]); return 2; }
    const std::string scenePath = argv[1], outPath = argv[2];
    int spp = 0, gpus = 1;
    float crop[4] = {0, 1, 0, 1}; bool haveCrop = false;
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--spp") && i + 1 < argc) spp = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--crop") && i + 4 < argc) { for (int k = 0; k < 4; ++k) crop[k] = (float)std::atof(argv[++i]); haveCrop = true; }
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (gpus < 1) gpus = 1;

    // pbrtParseFile / the baked container (core/api.cpp, csrc/scene_io.cpp)
    HprtModel *model = nullptr;
    const bool baked = scenePath.size() > 5 && scenePath.substr(scenePath.size() - 5) == ".hprt";
    if (baked) TRY(hprt_model_load(scenePath.c_str(), &model));
    else TRY(hprt_model_parse(scenePath.c_str(), nullptr, 0, &model));
    if (const char *w = hprt_model_warnings(model)) if (*w) std::fprintf(stderr, "%s\n", w);
    HprtRenderOptions opt;
    TRY(hprt_model_get_options(model, &opt));
    if (spp > 0) opt.spp = spp;
    if (haveCrop) std::memcpy(opt.crop, crop, sizeof(crop));

    // BVHAccel's constructor on the host (accelerators/bvh.cpp:181-250)
    HprtBvh *bvh = nullptr;
    TRY(hprt_bvh_build(model, &bvh));

    // croppedPixelBounds (core/film.cpp:56-60)
    const int x0 = (int)std::ceil((float)opt.xres * opt.crop[0]), x1 = (int)std::ceil((float)opt.xres * opt.crop[1]);
    const int y0 = (int)std::ceil((float)opt.yres * opt.crop[2]), y1 = (int)std::ceil((float)opt.yres * opt.crop[3]);
    const int W = x1 - x0, H = y1 - y0;
    if (W <= 0 || H <= 0) { std::fprintf(stderr, "empty film\n"); return 1; }
    const size_t nPix = (size_t)W * (size_t)H;

    // one scene per GPU; rank r renders tiles r, r + N, ...
    std::vector<HprtScene *> scenes((size_t)gpus, nullptr);
    for (int g = 0; g < gpus; ++g) TRY(hprt_scene_create_from_model(model, bvh, g, &scenes[(size_t)g]));
    // one host thread per GPU (the renders are independent; errors are thread-local in the library, so each thread keeps its own)
    std::vector<HprtRenderStats> stats((size_t)gpus);
    std::vector<int> rcs((size_t)gpus, HPRT_OK);
    std::vector<std::string> errs((size_t)gpus);
    std::vector<std::thread> workers;
    for (int g = 0; g < gpus; ++g)
        workers.emplace_back([&, g] {
            HprtRenderDesc desc; std::memset(&desc, 0, sizeof(desc));
            desc.opt = opt; desc.tile_begin = g; desc.tile_end = 0; desc.tile_stride = gpus;
            desc.flags = gpus > 1 ? HPRT_RENDER_EXPORT_FOREIGN : 0;
            rcs[(size_t)g] = hprt_render(scenes[(size_t)g], &desc, nullptr, nullptr, &stats[(size_t)g]);      // the library-owned film of this scene
            if (rcs[(size_t)g] != HPRT_OK) errs[(size_t)g] = hprt_last_error();
        });
    for (std::thread &t : workers) t.join();
    HprtRenderStats total; std::memset(&total, 0, sizeof(total));
    double seconds = 0;
    for (int g = 0; g < gpus; ++g) {
        if (rcs[(size_t)g] != HPRT_OK) { std::fprintf(stderr, "hprt_render on GPU %d failed (%d): %s\n", g, rcs[(size_t)g], errs[(size_t)g].c_str()); return 1; }
        const HprtRenderStats &st = stats[(size_t)g];
        total.camera_rays += st.camera_rays; total.rays += st.rays; total.shadow_rays += st.shadow_rays;
        if (st.render_seconds > seconds) seconds = st.render_seconds;
    }
    if (gpus > 1) TRY(hprt_film_gather_local(scenes.data(), nullptr, gpus, nPix, 0));      // Film::MergeFilmTile across GPUs (RCCL)

    // Film::WriteImage (core/film.cpp:266-303)
    std::vector<float> xyzw(4 * nPix), rgb(3 * nPix);
    TRY(hprt_film_read(scenes[0], xyzw.data(), nPix));
    TRY(hprt_film_resolve(xyzw.data(), nPix, opt.film_scale, rgb.data()));
    TRY(hprt_write_pfm(outPath.c_str(), rgb.data(), W, H));
    std::printf("%s: %dx%d, %d spp on %d GPU(s): %.3f s, %.1f Mrays/s (%llu camera + %llu path + %llu shadow rays) -> %s\n", hprt_version(), W, H,
                opt.spp, gpus, seconds, seconds > 0 ? (double)(total.rays + total.shadow_rays) / seconds * 1e-6 : 0.0,
                (unsigned long long)total.camera_rays, (unsigned long long)total.rays, (unsigned long long)total.shadow_rays, outPath.c_str());
    for (HprtScene *s : scenes) hprt_scene_destroy(s);
    hprt_bvh_destroy(bvh);
    hprt_model_destroy(model);
    return 0;
}
