// integrators/hprt_path.cpp in the reference tree — see hprt_path_integrator.h.  NOT compiled in this repository.
#include "integrators/hprt_path.h"

#include "accelerators/hprt.h"
#include "film.h"
#include "paramset.h"
#include "progressreporter.h"
#include "samplers/halton.h"
#include "scene.h"
#include "stats.h"

namespace pbrt {

STAT_COUNTER("Integrator/Camera rays traced", nCameraRays);           // the counters SamplerIntegrator::Render and Scene feed
STAT_COUNTER("Intersections/Regular ray intersection tests", nIntersectionTests);   // (core/integrator.cpp:48, core/scene.cpp:40-42)
STAT_COUNTER("Intersections/Shadow ray intersection tests", nShadowTests);

void HprtPathIntegrator::Render(const Scene &scene) {
    // the aggregate MakeScene() built (Scene::aggregate is private, core/scene.h:76-79: HprtAccel remembers the last one constructed)
    const HprtAccel *accel = HprtAccel::TopLevel();
    if (!accel) { Error("Integrator \"hprt-path\" needs Accelerator \"hprt\""); return; }
    HprtBridgeAccel *dev = accel->Device(scene.lights, frame.light_strategy);
    if (!dev) return;

    HprtRenderDesc desc; memset(&desc, 0, sizeof(desc));
    hprt_bridge_fill_options(&frame, &desc.opt);
    desc.tile_begin = tileBegin; desc.tile_end = 0; desc.tile_stride = tileStride;        // core/integrator.cpp:237-244's tile grid
    desc.flags = pixelStats ? HPRT_RENDER_PIXEL_STATS : 0;
    // pay the workspace allocation now, outside the span the fork times (Timings/Rendertime, core/integrator.cpp:242-351)
    if (hprt_scene_reserve(hprt_bridge_accel_scene(dev), &desc) != HPRT_OK) { Error("hprt: %s", hprt_last_error()); return; }

    Film *film = camera->film;
    // Film::pixels (core/film.h:85-93; private: the friend declaration of integration/README.md).  The fork's Pixel carries its
    // GeneralStats, so the bridge is given the real layout instead of assuming xyz[3] + filterWeightSum packed.
    Film::Pixel *pixels = film->pixels.get();
    const size_t xyzOffset = (size_t)((char *)&pixels[0].xyz[0] - (char *)&pixels[0]);
    const size_t weightOffset = (size_t)((char *)&pixels[0].filterWeightSum - (char *)&pixels[0]);
    HprtRenderStats st;
    {
        ProgressReporter reporter(1, "Rendering");
        if (hprt_bridge_render(dev, &desc, pixels, sizeof(Film::Pixel), xyzOffset, weightOffset, &st) != HPRT_OK) {
            Error("hprt: %s", hprt_last_error());
            return;
        }
        reporter.Update();
        reporter.Done();
    }
    nCameraRays += st.camera_rays; nIntersectionTests += st.rays; nShadowTests += st.shadow_rays;
    LOG(INFO) << "hprt: tile loop " << st.render_seconds << " s, " << (st.rays + st.shadow_rays) / st.render_seconds * 1e-6 << " Mrays/s";

    if (pixelStats) {
        // Pixel::stats (core/film.h:91): rays, primitiveIntersections[P], leafNodeTraversals[P], bvhTreeNodeTraversals[P] per pixel —
        // what filmTile->GetPixel(pixel).stats += ray.stats accumulates (core/integrator.cpp:327-328)
        const Bounds2i b = film->croppedPixelBounds;
        const size_t n = (size_t)b.Area();
        std::vector<uint64_t> s7(7 * n);
        if (hprt_pixel_stats_read(hprt_bridge_accel_scene(dev), s7.data(), n) == HPRT_OK)
            for (size_t i = 0; i < n; ++i) {
                GeneralStats &g = pixels[i].stats;                     // core/geometry.h:1078-1173
                g.rays = s7[7 * i]; g.primitiveIntersections = s7[7 * i + 1]; g.primitiveIntersectionsP = s7[7 * i + 2];
                g.leafNodeTraversals = s7[7 * i + 3]; g.leafNodeTraversalsP = s7[7 * i + 4];
                g.bvhTreeNodeTraversals = s7[7 * i + 5]; g.bvhTreeNodeTraversalsP = s7[7 * i + 6];
            }
    }
    film->WriteImage();                                                // core/integrator.cpp:358
    film->WriteGeneralStats();                                         // :359 — the fork's heat-map text files
}

HprtPathIntegrator *CreateHprtPathIntegrator(const ParamSet &params, const ParamSet &filmParams, const ParamSet &cameraParams,
                                             const ParamSet &acceleratorParams, std::shared_ptr<Sampler> sampler,
                                             std::shared_ptr<const Camera> camera) {
    HprtBridgeFrame f; memset(&f, 0, sizeof(f));
    const Film *film = camera->film;
    // ---- PathIntegrator's parameters (integrators/path.cpp:206-229) ----
    f.max_depth = params.FindOneInt("maxdepth", 5);
    f.rr_threshold = params.FindOneFloat("rrthreshold", 1.);
    const std::string strategy = params.FindOneString("lightsamplestrategy", "spatial");
    f.light_strategy = strategy == "uniform" ? 0 : strategy == "power" ? 1 : 2;          // core/lightdistrib.cpp:47-66
    int np; if (params.FindInt("pixelbounds", &np)) Warning("hprt-path: \"pixelbounds\" is outside the hot path's scope; use the film's cropwindow");
    // ---- Film (core/film.cpp:310-351: CreateFilm) ----
    f.full_resolution[0] = film->fullResolution.x; f.full_resolution[1] = film->fullResolution.y;
    f.crop_window[0] = 0; f.crop_window[1] = 1; f.crop_window[2] = 0; f.crop_window[3] = 1;
    int cwi; const Float *cr = filmParams.FindFloat("cropwindow", &cwi);
    if (cr && cwi == 4) {                                              // core/film.cpp:326-341, PbrtOptions.cropWindow aside
        f.crop_window[0] = Clamp(std::min(cr[0], cr[1]), 0.f, 1.f); f.crop_window[1] = Clamp(std::max(cr[0], cr[1]), 0.f, 1.f);
        f.crop_window[2] = Clamp(std::min(cr[2], cr[3]), 0.f, 1.f); f.crop_window[3] = Clamp(std::max(cr[2], cr[3]), 0.f, 1.f);
    }
    f.filter_radius[0] = film->filter->radius.x; f.filter_radius[1] = film->filter->radius.y;   // box, 0.5 x 0.5: anything else is refused by hprt_render
    f.film_scale = filmParams.FindOneFloat("scale", 1.);
    f.max_sample_luminance = filmParams.FindOneFloat("maxsampleluminance", Infinity);
    // ---- PerspectiveCamera (cameras/perspective.cpp:224-271) ----
    if (!dynamic_cast<const PerspectiveCamera *>(camera.get())) Error("hprt-path: only the perspective camera is in the hot path's scope");
    const Matrix4x4 &c2w = camera->CameraToWorld.startTransform->GetMatrix(), &w2c = camera->CameraToWorld.startTransform->GetInverseMatrix();
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { f.camera_to_world[4 * i + j] = c2w.m[i][j]; f.world_to_camera[4 * i + j] = w2c.m[i][j]; }
    f.lens_radius = cameraParams.FindOneFloat("lensradius", 0.f);
    f.focal_distance = cameraParams.FindOneFloat("focaldistance", 1e6);
    const Float frame = cameraParams.FindOneFloat("frameaspectratio", Float(film->fullResolution.x) / Float(film->fullResolution.y));
    if (frame > 1.f) { f.screen_window[0] = -frame; f.screen_window[1] = frame; f.screen_window[2] = -1.f; f.screen_window[3] = 1.f; }
    else { f.screen_window[0] = -1.f; f.screen_window[1] = 1.f; f.screen_window[2] = -1.f / frame; f.screen_window[3] = 1.f / frame; }
    int swi; const Float *sw = cameraParams.FindFloat("screenwindow", &swi);
    if (sw && swi == 4) for (int i = 0; i < 4; ++i) f.screen_window[i] = sw[i];
    f.fov = cameraParams.FindOneFloat("fov", 90.);
    const Float halffov = cameraParams.FindOneFloat("halffov", -1.f);
    if (halffov > 0.f) f.fov = 2.f * halffov;
    // ---- HaltonSampler (samplers/halton.cpp:133-142) ----
    const HaltonSampler *halton = dynamic_cast<const HaltonSampler *>(sampler.get());
    if (!halton) Error("hprt-path: only Sampler \"halton\" is in the hot path's scope (every bundled scene uses it)");
    f.samples_per_pixel = (int32_t)sampler->samplesPerPixel;
    f.sample_at_pixel_center = halton && halton->sampleAtPixelCenter ? 1 : 0;
    // ---- the aggregate's build parameters (accelerators/bvh.cpp:529-535) ----
    f.max_node_prims = acceleratorParams.FindOneInt("maxnodeprims", 4);
    f.isect_cost = acceleratorParams.FindOneInt("intersectcost", 8);
    f.trav_cost = acceleratorParams.FindOneInt("traversalcost", 1);
    // multi-process runs (scripts/run_distributed.sh's machines, or one process per GPU): HPRT_RANK / HPRT_WORLD deal the tiles
    // round-robin; the films are merged by hprt_film_gather
    const char *r = getenv("HPRT_RANK"), *w = getenv("HPRT_WORLD");
    const int world = w ? std::max(1, atoi(w)) : 1, rank = r ? std::min(world - 1, std::max(0, atoi(r))) : 0;
    return new HprtPathIntegrator(f, camera, params.FindOneBool("pixelstats", true), rank, world);
}

}  // namespace pbrt
