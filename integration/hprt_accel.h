// accelerators/hprt.h in the reference tree — Accelerator "hprt": pbrt's Aggregate seam over libhprt.so.
//
// Written against the reference's headers (jhoobergs/Thesis-pbrt-v3, src/); it is NOT compiled in this repository (the
// reference's glog / OpenEXR / Ptex submodules are empty here and stand-in headers are not allowed).  Everything that touches
// include/hprt.h lives in integration/hprt_bridge.cpp, which IS compiled and tested (tests/test_integration_bridge.py); this
// class only copies pbrt members into the bridge's plain structs.  integration/README.md lists the one-line patches the
// reference needs (factory branches, friend declarations, CMake).
#ifndef PBRT_ACCELERATORS_HPRT_H
#define PBRT_ACCELERATORS_HPRT_H

#include "pbrt.h"
#include "primitive.h"
#include "hprt_bridge.h"

namespace pbrt {

class HprtAccel : public Aggregate {                                  // core/primitive.h:185-196
  public:
    HprtAccel(std::vector<std::shared_ptr<Primitive>> p, int maxPrimsInNode, int isectCost, int travCost);
    ~HprtAccel();
    Bounds3f WorldBound() const { return bounds; }                    // accelerators/bvh.cpp:187-189
    // The single-ray contract of Aggregate (core/primitive.h:57-61), met through the batch ABI with n = 1: a host-device round
    // trip per ray.  It exists so that code outside the integrator that asks the scene a question (a light's Preprocess, a
    // debugging tool) gets the reference's answer; HprtPathIntegrator never comes through here.
    bool Intersect(const Ray &ray, SurfaceInteraction *isect) const;
    bool IntersectP(const Ray &ray) const;

    // ---- for HprtPathIntegrator ----
    // The device scene, created on first use: an HprtAccel is also what pbrtObjectInstance builds over an object definition's
    // primitives (core/api.cpp:1798-1806 calls MakeAccelerator with the scene's accelerator name); those never reach the device
    // on their own — the top-level aggregate finds them behind its TransformedPrimitives and uploads them as HprtObjectDescs.
    HprtBridgeAccel *Device(const std::vector<std::shared_ptr<Light>> &lights, int lightStrategy) const;
    // the aggregate MakeScene() built last: Scene::aggregate is private in the reference (core/scene.h:76-79)
    static const HprtAccel *TopLevel() { return topLevel; }

  private:
    friend class HprtSceneWalk;
    std::vector<std::shared_ptr<Primitive>> primitives;               // in the order given: BVHAccel keeps the same (bvh.h:69)
    const int maxPrimsInNode, isectCost, travCost;
    Bounds3f bounds;
    mutable HprtBridgeAccel *device = nullptr;
    mutable std::mutex deviceMutex;
    static const HprtAccel *topLevel;
};

std::shared_ptr<HprtAccel> CreateHprtAccelerator(std::vector<std::shared_ptr<Primitive>> prims, const ParamSet &ps);

}  // namespace pbrt

#endif  // PBRT_ACCELERATORS_HPRT_H
