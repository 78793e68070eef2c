// integrators/hprt_path.h in the reference tree — Integrator "hprt-path": pbrt's Integrator seam over libhprt.so.
// Written against the reference's headers; NOT compiled in this repository (see hprt_accel.h).  All calls into include/hprt.h go
// through integration/hprt_bridge.cpp, which is compiled and tested.
#ifndef PBRT_INTEGRATORS_HPRT_PATH_H
#define PBRT_INTEGRATORS_HPRT_PATH_H

#include "pbrt.h"
#include "camera.h"
#include "integrator.h"
#include "hprt_bridge.h"

namespace pbrt {

// SamplerIntegrator::Render (core/integrator.cpp:230-360) with PathIntegrator::Li (integrators/path.cpp:64-204) as one call:
// same tiles, same Halton sample per (pixel, sample index, dimension), same film — computed by the wavefront kernels.
class HprtPathIntegrator : public Integrator {                       // core/integrator.h:53-58
  public:
    HprtPathIntegrator(const HprtBridgeFrame &frame, std::shared_ptr<const Camera> camera, bool pixelStats, int tileBegin, int tileStride)
        : frame(frame), camera(camera), pixelStats(pixelStats), tileBegin(tileBegin), tileStride(tileStride) {}
    void Render(const Scene &scene) override;

  private:
    const HprtBridgeFrame frame;
    std::shared_ptr<const Camera> camera;                             // its film receives the result (core/camera.h:68)
    const bool pixelStats;                                            // keep the fork's per-pixel GeneralStats (core/film.h:91)
    const int tileBegin, tileStride;                                  // this process's share of the 16x16 tile grid (multi-process runs)
};

// MakeIntegrator's branch (core/api.cpp:1914-1915) calls this with the parameter sets RenderOptions holds: the integrator's own, and
// the film's and camera's for the values the constructed Film / PerspectiveCamera no longer expose (crop window, fov, screen window)
HprtPathIntegrator *CreateHprtPathIntegrator(const ParamSet &integratorParams, const ParamSet &filmParams, const ParamSet &cameraParams,
                                             const ParamSet &acceleratorParams, std::shared_ptr<Sampler> sampler,
                                             std::shared_ptr<const Camera> camera);

}  // namespace pbrt

#endif  // PBRT_INTEGRATORS_HPRT_PATH_H
