// hprt — the four-wide walk record (plain data, shared by the host builder wide_bvh.cpp and the kernels; see wide_bvh.h).
#pragma once
#include <cstdint>

namespace hprt {

// word 0: grid origin x y z, em = ex | ey << 8 | ez << 16 | meta << 24 (e*: biased float exponent of the grid step;
//         meta: split axis of the collapsed node, bits 0-1, of its first child, bits 2-3, of its second child, bits 4-5)
// word 1: qlo.x qhi.x qlo.y qhi.y   word 2: qlo.z qhi.z - -     (byte s of a word: slot s)
// word 3: ref[4]: >= 0 wide node, WIDE_NONE empty slot, otherwise a leaf (below)
struct DevWide { float o[3]; uint32_t em; uint32_t q[6]; uint32_t pad[2]; int32_t ref[4]; };
static_assert(sizeof(DevWide) == 64, "DevWide must be 64 bytes");

// Leaf references: ~firstPrimitive (ordered index < 2^28) with two state bits.
//   bits 30,29 = 1,1  entry of a leaf that holds exactly one triangle: its exact box is the min / max of the vertices
//   bit 30 = 0        entry of any other leaf: its exact box is read from DevScene::leafBox
//   bits 30,29 = 1,0  a later primitive of a leaf whose box test has passed
enum : uint32_t { WIDE_LEAF_BOXED = 0x40000000u, WIDE_LEAF_FIRST = 0x20000000u };
enum : int32_t { WIDE_NONE = (int32_t)0x80000000 };

}  // namespace hprt
