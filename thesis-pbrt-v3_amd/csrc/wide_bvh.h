// hprt — four-wide walk structure over the reference's binary BVH ("leaf-exact walk").
//
// Bounds3::IntersectP (core/geometry.h:1754-1780) is monotone under box inclusion, and an interior node's bounds are
// exactly the Union of the primitive bounds below it (accelerators/bvh.cpp:220-222): a node whose box passes the test has
// only ancestors whose boxes pass.  The walk of BVHAccel::Intersect / IntersectP (accelerators/bvh.cpp:354-437) therefore
// reaches a leaf if and only if the leaf's OWN box passes (against the tMax current when the leaf comes up), and what the
// walk returns is decided by (1) the order in which leaves come up, (2) each leaf's own exact box test and (3) the
// primitive tests.  The interior boxes only have to be conservative: never cull what the exact test would pass.
//
// DevWide collapses two levels of the reference's tree into one record of 64 bytes (four 16-byte requests, as a DevPair,
// but deciding up to four grandchildren): child boxes quantised to 8 bits per coordinate on a per-node grid (origin +
// q * 2^e), rounded OUTWARDS in the arithmetic the kernel dequantises with, so that a dequantised box always contains the
// exact one and, by the monotonicity above, passes whenever the exact box does.  Slots 0-1 are the children of the first
// child (or the first child itself in slot 0 when it is a leaf), slots 2-3 those of the second child: with the three
// split axes of `meta` the kernel visits the slots in the order the reference's nested near/far decisions give.
#pragma once
#include <cstdint>
#include <vector>
#include "bvh_builder.h"
#include "device/dev_wide.h"

namespace hprt {

// nd: the reference's linear node array of one aggregate; leafRef[i]: the reference (as above) of leaf node i.
// Appends the aggregate's wide nodes to *out (child indices are absolute: they include out->size() at entry); *stackNeed
// receives the largest number of pending entries a walk can hold.  false: some box is not finite or its extent does not
// fit the grid — the caller keeps the binary walk for the scene.
bool BuildWide(const BvhNode *nd, uint32_t nNodes, const int32_t *leafRef, std::vector<DevWide> *out, int *stackNeed);

}  // namespace hprt
