// octree_ref.h -- TEST INFRASTRUCTURE ONLY (part of the CPU checker).
// Interface of the CPU restatement of FastOctTree<Voxel>'s query side
// (cpu_rt/src/ray_tracer/hittable/fast_oct_tree/{mod,ray_trace,arena}.rs).
#ifndef ORACLE_OCTREE_REF_H
#define ORACLE_OCTREE_REF_H

#include <cstdint>
#include <string>
#include <vector>

#include "../include/mrt.h"

namespace octref {

struct Ray {
    double origin[3];
    double direction[3];
    double time;
};
struct HitInfo {
    double hit_position[3];
    double normal[3];
    int voxel;   // index into the tree's voxel table
    bool solid;  // HitInfo::Solid vs HitInfo::Volume
};

struct Tree {
    bool from_linear(const mrt_octree&, std::string* e) {
        *e = "octree shapes are not supported by this build of the checker";
        return false;
    }
    uint32_t world_size() const { return 0; }
    bool is_empty() const { return true; }
    bool trace_ray(const Ray&, HitInfo*) const { return false; }
};

}  // namespace octref
#endif
