// hprt — library-internal glue shared by capi_host.cpp and capi_device.hip.
#pragma once
#include <string>
#include <vector>
#include "bvh_builder.h"
#include "scene_model.h"

struct HprtModel { hprt::SceneModel sc; };
struct HprtBvh {
    hprt::BvhTree tree;                       // top-level aggregate (renderOptions->primitives)
    std::vector<hprt::BvhTree> objects;       // one per object definition (the accelerator ObjectInstance builds, core/api.cpp:1798-1806)
};

namespace hprt {
extern thread_local std::string g_lastError;
int SetError(int code, const std::string &msg);
int HandleException();      // maps the exception in flight to an HPRT_E_* code + message (capi_host.cpp)
}  // namespace hprt
