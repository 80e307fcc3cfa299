// rt_bvh_build.h — on-device construction of the traversal BVH (rt_bvh_build.hip), used by rt_scene_upload.
#ifndef RT_BVH_BUILD_H
#define RT_BVH_BUILD_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace rtb {

struct Result {
    float *nodes = nullptr;       // n_nodes x RT_NODE_FLOATS, device memory owned by the caller (hipFree); null if the root is a leaf
    uint32_t n_nodes = 0;
    int32_t root_ref = 0;
    uint32_t stack_depth = 1;
    uint32_t *perm = nullptr;     // n entries, device memory owned by the caller: slot i of the new order = primitive perm[i]
    float build_ms = 0.0f;        // Morton codes .. quantised nodes, measured with HIP events on `stream`
    uint32_t rounds = 0;          // clustering rounds (0 for the radix tree)
};

// boxes: n x 6 floats (lo xyz, hi xyz) in device memory; lo / hi: bounds of the box CENTRES.
// Returns an rt_status; on failure *err has the text and whatever *out holds must still be freed.
// algo: RT_BVH_DEVICE (clustering) or RT_BVH_DEVICE_LBVH (Morton radix tree), include/rt_api.h.
int build_bvh(hipStream_t stream, const float *d_boxes, uint32_t n, uint32_t leaf_max, uint32_t algo, const float lo[3], const float hi[3],
              Result *out, std::string *err);

// dst[i] = src[perm[i]] for records of floats_per_record floats (a multiple of 4)
void gather_records(hipStream_t stream, const void *src, const uint32_t *perm, uint32_t n, uint32_t floats_per_record, void *dst);
void invert_permutation(hipStream_t stream, const uint32_t *perm, uint32_t n, uint32_t *inv);
// list[i] = inv[list[i]] for entries below n (slots that took part in the build)
void remap_slot_list(hipStream_t stream, uint32_t *list, uint32_t count, const uint32_t *inv, uint32_t n);

}  // namespace rtb
#endif
