// hprt — library-internal glue shared by capi_host.cpp and capi_device.hip.
#pragma once
#include <string>
#include "bvh_builder.h"
#include "scene_model.h"

struct HprtModel { hprt::SceneModel sc; };
struct HprtBvh { hprt::BvhTree tree; };

namespace hprt {
extern thread_local std::string g_lastError;
int SetError(int code, const std::string &msg);
}  // namespace hprt
