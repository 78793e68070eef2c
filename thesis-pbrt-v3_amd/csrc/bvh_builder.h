// hprt host side — builder for the 32-byte flattened BVH node array that the
// traversal kernels consume.  The array is byte-identical to what the pbrt-v3
// fork's BVHAccel constructor produces (accelerators/bvh.cpp:123-152 layout,
// :196-333 split rule, :335-350 depth-first flattening), because closest-hit tie
// breaking depends on node order, near/far order and in-leaf order.
#pragma once
#include <cstdint>
#include <vector>
#include "scene_model.h"

namespace hprt {

struct BvhNode {                 // == LinearBVHNode, accelerators/bvh.cpp:123-152
    float bmin[3], bmax[3];
    int32_t offset;              // leaf: first primitive (ordered list); interior: second child
    uint32_t countAxis;          // (nPrimitives << 2) | axis, axis == 3 marks a leaf
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 32 bytes");

struct BvhTree {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> primOrder;   // ordered position -> creation-order primitive number
    int maxDepth = 0, nLeaves = 0;
};

// World bounds of every primitive of an aggregate in creation order (Triangle::WorldBound,
// shapes/triangle.cpp:180-186; Shape::WorldBound for spheres, core/shape.cpp:53;
// TransformedPrimitive::WorldBound for instances, which needs the objects' trees).
void ComputeObjectPrimBounds(const SceneModel &sc, int object, std::vector<float> *bmin, std::vector<float> *bmax);
void ComputePrimBounds(const SceneModel &sc, const std::vector<BvhTree> &objectTrees, std::vector<float> *bmin,
                       std::vector<float> *bmax);

// Full-sweep SAH build.  bmin/bmax: 3 floats per primitive.
void BuildBvh(size_t nPrims, const float *bmin, const float *bmax, int maxPrimsInNode, int isectCost, int travCost,
              BvhTree *out);

}  // namespace hprt
