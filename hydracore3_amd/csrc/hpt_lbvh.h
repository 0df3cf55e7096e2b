// Interface of the device-side CommitScene (hpt_lbvh.hip): a linear BVH over the instanced triangles of the single-level layout and its 4-wide
// compressed form, built by kernels. Called by hpt_host.hip's commit path when the caller asks for a fast build (CommitScene(BUILD_LOW / BUILD_MEDIUM),
// CrossRT.h:8-14, 109) or hpt_set_option("device_build", 1).
#pragma once
#include <string>
#include <vector>
#include "hpt_types.h"

namespace hpt {

struct LbvhInstance { float objectToWorld[12]; uint triCount; size_t posOffset, idxOffset; };   // rows of the 3x4 matrix; offsets (floats / uints) of its mesh in the uploaded geometry
struct LbvhResult { uint rootRef = REF_NONE, numNodes = 0, depth = 0, nodes4Count = 0, depth4 = 0; float sahVisits = 1.0f; };

void* lbvhCreate();
void  lbvhDestroy(void* scratch);
// copies every mesh's positions (3 floats per vertex) and indices to the device; posOffset / idxOffset receive each mesh's place
bool  lbvhUploadGeometry(void* scratch, const std::vector<const float*>& pos, const std::vector<size_t>& posFloats, const std::vector<const uint*>& idx, const std::vector<size_t>& idxCount,
                         std::vector<size_t>& posOffset, std::vector<size_t>& idxOffset, std::string& err);
// builds into caller-owned device arrays: nodes (n - 1 entries), tris (n), nodes4 (cap4 entries; skipped when !wantWide)
bool  lbvhBuild(void* scratch, const LbvhInstance* insts, uint ni, uint n, BvhNode* nodes, BvhTri* tris, BvhNode4* nodes4, uint cap4, bool wantWide, LbvhResult& out, std::string& err);

} // namespace hpt
