// octree_ref.h -- TEST INFRASTRUCTURE ONLY (part of the CPU checker).
//
// CPU restatement of FastOctTree (cpu_rt/src/ray_tracer/hittable/fast_oct_tree/): the
// generational arena (arena.rs), Node::{set_child, get, pos_to_index, world_pos_to_child_pos,
// world_pos_to_child_index} (mod.rs:35-229), FastOctTree::{set, get, combine, world_size}
// (operations/set.rs, operations/combine.rs, mod.rs:236-263), get_chunk / get_chunk_size /
// get_step_size / in_range (ray_trace.rs:15-83) and trace_ray / ray_iteration for solid voxels
// (ray_trace.rs:86-698). Leaves carry an integer payload (the reference is generic over the leaf
// type; its tests use u32 and Voxel leaves); equality of payloads stands for equality of voxels.
// Volume voxels (handle_volume / handle_empty, ray_trace.rs:118-199) are walked when a voxel table is
// given (VolumeCtx); without one every leaf is a solid voxel (the reference's u32 test trees).
//
// Pinned by the reference's own unit tests, restated in tests/test_octree.py.
#ifndef ORACLE_OCTREE_REF_H
#define ORACLE_OCTREE_REF_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../include/mrt.h"

namespace octref {

struct ArenaIndex {
    size_t index;
    uint32_t generation;
};
enum class Kind { Parent, Leaf, Empty };
struct Node {
    Kind kind = Kind::Empty;
    ArenaIndex children[8]{};
    int32_t leaf = 0;
    uint32_t size = 0;  // log2 of the edge
    bool data_equal(const Node& o) const {
        if (kind != o.kind) return false;
        if (kind == Kind::Leaf) return leaf == o.leaf;
        if (kind == Kind::Parent) {
            for (int i = 0; i < 8; ++i)
                if (children[i].index != o.children[i].index || children[i].generation != o.children[i].generation)
                    return false;
        }
        return true;
    }
    bool operator==(const Node& o) const { return data_equal(o) && size == o.size; }  // #[derive(PartialEq)]
};

// arena.rs:22-117
struct Arena {
    struct Slot {
        Node data;
        uint32_t generation;
    };
    std::vector<Slot> data;
    std::set<size_t> deleted_indices;  // BTreeSet: pop_first takes the lowest free slot

    ArenaIndex insert(const Node& n) {
        if (!deleted_indices.empty()) {
            size_t index = *deleted_indices.begin();
            deleted_indices.erase(deleted_indices.begin());
            data[index].data = n;
            data[index].generation += 1;
            return {index, data[index].generation};
        }
        size_t index = data.size();
        data.push_back({n, 0});
        return {index, 0};
    }
    bool key_exists(ArenaIndex i) const {
        if (i.index >= data.size() || deleted_indices.count(i.index)) return false;
        return i.generation == data[i.index].generation;
    }
    const Node* get(ArenaIndex i) const { return key_exists(i) ? &data[i.index].data : nullptr; }
    const Node& get_unchecked(ArenaIndex i) const { return data[i.index].data; }
    bool update(ArenaIndex i, const Node& n) {
        if (!key_exists(i)) return false;  // assert!
        data[i.index].data = n;
        return true;
    }
    const Node* get_root() const { return data.empty() ? nullptr : &data[0].data; }
    void update_root(const Node& n) {
        if (!data.empty()) {
            uint32_t g = data[0].generation + 1;
            data[0] = {n, g};
        } else {
            data.push_back({n, 0});
        }
    }
    bool del(ArenaIndex i) {
        if (!key_exists(i)) return false;  // assert!
        deleted_indices.insert(i.index);
        return true;
    }
};

struct P3u {
    uint32_t x, y, z;
};
struct P3i {
    int32_t x, y, z;
};

// mod.rs:195-225
inline int pos_to_index(P3u p) {
    if (p.x > 1 || p.y > 1 || p.z > 1) return -1;  // panic!("unsupported position")
    return (int)(p.x * 4 + p.y * 2 + p.z);
}
inline P3u world_pos_to_child_pos(P3u p, uint32_t size) {
    uint32_t mask = ~(1u << (size - 1));
    return {p.x & mask, p.y & mask, p.z & mask};
}
inline int world_pos_to_child_index(P3u p, uint32_t size) {
    uint32_t cs = size - 1;
    return pos_to_index({p.x >> cs, p.y >> cs, p.z >> cs});
}
// ray_trace.rs:686-698
inline int32_t floor_size(int32_t value, uint32_t size) { return (value >> size) << size; }
inline int32_t int_sign(double a) {
    uint64_t bits;
    std::memcpy(&bits, &a, 8);
    return ((~bits) >> 63) == 1 ? 1 : -1;
}
inline int32_t f64_as_i32(double v) {  // Rust `as i32`: truncation toward zero, saturating, NaN -> 0
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (int32_t)0x80000000;
    return (int32_t)v;
}

struct Ray {
    double origin[3];
    double direction[3];
    double time;
};
// HitInfo<Voxel> (hit_info.rs): Solid { hit_value, hit_position, normal } or Volume { hit_value,
// hit_position }. `voxel` indexes the voxel table; `edge` = the hit material is the solid
// material of that volume voxel's edge effect (ray_trace.rs:155-166), not the voxel's own.
struct HitInfo {
    double hit_position[3];
    double normal[3];
    int voxel;
    bool solid;
    bool edge;
};
// Voxel table and random source for volume voxels. rand01(user, 2 * n + which): the uniform of the
// n-th volume entry of this walk; which = 0 edge effect (ray_trace.rs:159), 1 free path (:168).
// Steps of ray_iteration made by the calling thread (a diagnostic for walk-length statistics).
inline unsigned long long& oct_step_counter() {
    static thread_local unsigned long long n = 0;
    return n;
}
struct VolumeCtx {
    const mrt_voxel* voxels = nullptr;
    double (*rand01)(void* user, uint32_t index) = nullptr;
    void* user = nullptr;
    uint32_t* entry = nullptr;  // running count of volume entries (n above)
};

struct Tree {
    Arena arena;

    uint32_t world_size() const {  // mod.rs:243-249
        const Node* r = arena.get_root();
        return r ? (1u << r->size) : 0u;
    }
    bool is_empty() const { return world_size() == 0; }

    // Node::set_child, mod.rs:43-169
    Node set_child(Node self, int32_t value, P3u position) {
        if (self.size == 0) {
            self.kind = Kind::Leaf;
            self.leaf = value;
            return self;
        }
        if (self.kind == Kind::Empty) {
            Node child;
            child.kind = Kind::Empty;
            child.size = self.size - 1;
            for (int i = 0; i < 8; ++i) self.children[i] = arena.insert(child);
            self.kind = Kind::Parent;
        } else if (self.kind == Kind::Leaf) {
            Node child;
            child.kind = Kind::Leaf;
            child.leaf = self.leaf;
            child.size = self.size - 1;
            for (int i = 0; i < 8; ++i) self.children[i] = arena.insert(child);
            self.kind = Kind::Parent;
        }
        int index = world_pos_to_child_index(position, self.size);
        P3u pos_in_child = world_pos_to_child_pos(position, self.size);
        Node child = *arena.get(self.children[index]);
        Node child_clone = set_child(child, value, pos_in_child);
        arena.update(self.children[index], child_clone);
        Node c[8];
        for (int i = 0; i < 8; ++i) c[i] = *arena.get(self.children[i]);
        bool all_equal = true;
        for (int i = 1; i < 8; ++i) all_equal = all_equal && (c[0] == c[i]);
        if (all_equal && (c[0].kind == Kind::Leaf || c[0].kind == Kind::Empty)) {
            ArenaIndex kids[8];
            for (int i = 0; i < 8; ++i) kids[i] = self.children[i];
            self.kind = c[0].kind;
            self.leaf = c[0].leaf;
            for (int i = 0; i < 8; ++i) arena.del(kids[i]);
        }
        return self;
    }
    // operations/set.rs:4-36
    void set(int32_t value, P3u position) {
        const Node* rootp = arena.get_root();
        if (rootp) {
            Node root = *rootp;
            uint32_t ws = 1u << root.size;
            if (position.x < ws && position.y < ws && position.z < ws) {
                Node nr = set_child(root, value, position);
                arena.update_root(nr);
            } else {
                uint32_t old_size = root.size;
                ArenaIndex old_root = arena.insert(root);
                Node nr;
                nr.kind = Kind::Parent;
                nr.size = old_size + 1;
                nr.children[0] = old_root;
                Node e;
                e.kind = Kind::Empty;
                e.size = old_size;
                for (int i = 1; i < 8; ++i) nr.children[i] = arena.insert(e);
                arena.update_root(nr);
                set(value, position);
            }
        } else {
            Node e;
            arena.insert(e);
            set(value, position);
        }
    }
    // Node::get, mod.rs:171-184; FastOctTree::get, mod.rs:251-262. Returns false for None.
    bool get(P3u position, int32_t* out) const {
        const Node* r = arena.get_root();
        if (!r) return false;
        uint32_t ws = 1u << r->size;
        if (!(position.x < ws && position.y < ws && position.z < ws)) return false;
        Node cur = *r;
        for (;;) {
            if (cur.kind == Kind::Empty) return false;
            if (cur.kind == Kind::Leaf) {
                *out = cur.leaf;
                return true;
            }
            int ci = world_pos_to_child_index(position, cur.size);
            position = world_pos_to_child_pos(position, cur.size);
            cur = *arena.get(cur.children[ci]);
        }
    }
    // operations/combine.rs:4-19
    void combine(const Tree& other, P3i offset) {
        uint32_t ws = other.world_size();
        for (uint32_t x = 0; x < ws; ++x)
            for (uint32_t y = 0; y < ws; ++y)
                for (uint32_t z = 0; z < ws; ++z) {
                    int32_t sx = (int32_t)x + offset.x, sy = (int32_t)y + offset.y, sz = (int32_t)z + offset.z;
                    if (sx >= 0 && sy >= 0 && sz >= 0) {
                        int32_t v;
                        if (other.get({x, y, z}, &v)) set(v, {(uint32_t)sx, (uint32_t)sy, (uint32_t)sz});
                    }
                }
    }

    // ray_trace.rs:15-19
    bool in_range(P3i p) const {
        int32_t ws = (int32_t)world_size();
        return p.x >= 0 && p.x < ws && p.y >= 0 && p.y < ws && p.z >= 0 && p.z < ws;
    }
    // ray_trace.rs:23-51; returns -1 where the reference panics (out of range)
    int get_chunk_size(P3i position) const {
        if (!in_range(position)) return -1;
        Node cur = *arena.get_root();
        P3u p{(uint32_t)position.x, (uint32_t)position.y, (uint32_t)position.z};
        for (;;) {
            if (cur.kind != Kind::Parent) return (int)cur.size;
            int ci = world_pos_to_child_index(p, cur.size);
            p = world_pos_to_child_pos(p, cur.size);
            cur = arena.get_unchecked(cur.children[ci]);
        }
    }
    int get_step_size(P3i position) const {  // ray_trace.rs:20-22
        int s = get_chunk_size(position);
        return s < 0 ? -1 : (1 << s);
    }
    // ray_trace.rs:55-83; returns false for None
    bool get_chunk(P3u position, Node* out) const {
        if (!in_range({(int32_t)position.x, (int32_t)position.y, (int32_t)position.z})) return false;
        const Node* r = arena.get_root();
        if (!r) return false;
        Node cur = *r;
        P3u p = position;
        for (;;) {
            if (cur.kind != Kind::Parent) {
                *out = cur;
                return true;
            }
            int ci = world_pos_to_child_index(p, cur.size);
            p = world_pos_to_child_pos(p, cur.size);
            cur = arena.get_unchecked(cur.children[ci]);
        }
    }

    // RayTraceState::volume_distance_left (ray_trace.rs:93-104)
    struct VolumeState {
        bool some = false;
        double distance_left = 0.0, first_density = 0.0;
        double last_position[3] = {0, 0, 0};
        int previous = -1;  // previous_volume (index into the voxel table)
    };
    enum class Step { Continue, StopVolume, StopSolid };
    static double dist3(const double* a, const double* b) {  // cgmath distance = (a - b).magnitude()
        double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
        return std::sqrt(x * x + y * y + z * z);
    }
    static void stop_position(const VolumeState& vs, const double* d, double* out) {  // last + left * d.normalize()
        double inv = 1.0 / std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int a = 0; a < 3; ++a) out[a] = vs.last_position[a] + (d[a] * inv) * vs.distance_left;
    }
    // handle_volume, ray_trace.rs:118-179
    static Step handle_volume(int v, VolumeState& vs, const double* cur, const double* d, const double* normal,
                              const VolumeCtx& ctx, HitInfo* out) {
        const mrt_voxel& vm = ctx.voxels[v];
        if (vs.some) {
            double previous_density = ctx.voxels[vs.previous].density;
            double density_ratio = vs.first_density / previous_density;
            double distance_traveled = dist3(cur, vs.last_position) * density_ratio;
            if (distance_traveled > vs.distance_left) {
                stop_position(vs, d, out->hit_position);
                out->solid = false, out->edge = false, out->voxel = vs.previous;
                out->normal[0] = 1.0, out->normal[1] = 0.0, out->normal[2] = 0.0;
                return Step::StopVolume;
            }
            vs.distance_left = vs.distance_left - distance_traveled;
            for (int a = 0; a < 3; ++a) vs.last_position[a] = cur[a];
            return Step::Continue;
        }
        const uint32_t n = (*ctx.entry)++;
        if (vm.edge_kind != MRT_VOX_EDGE_NONE) {
            float random_number = (float)ctx.rand01(ctx.user, 2 * n);
            if (random_number < vm.edge_probability) {
                out->solid = true, out->edge = true, out->voxel = v;
                for (int a = 0; a < 3; ++a) out->hit_position[a] = cur[a], out->normal[a] = normal[a];
                return Step::StopSolid;
            }
        }
        vs.some = true;
        vs.distance_left = std::log(ctx.rand01(ctx.user, 2 * n + 1)) / (-vm.density);
        for (int a = 0; a < 3; ++a) vs.last_position[a] = cur[a];
        vs.first_density = vm.density;
        vs.previous = v;
        return Step::Continue;
    }
    // handle_empty, ray_trace.rs:180-199
    static Step handle_empty(VolumeState& vs, const double* cur, const double* d, HitInfo* out) {
        if (!vs.some) return Step::Continue;
        double distance_traveled = dist3(cur, vs.last_position);
        if (distance_traveled > vs.distance_left) {
            stop_position(vs, d, out->hit_position);
            out->solid = false, out->edge = false, out->voxel = vs.previous;
            out->normal[0] = 1.0, out->normal[1] = 0.0, out->normal[2] = 0.0;
            return Step::StopVolume;
        }
        vs.some = false;
        return Step::Continue;
    }

    // ray_iteration, ray_trace.rs:86-560. Returns false for None.
    bool ray_iteration(P3i block_coordinates, const Ray& ray, const double* initial_normal, const VolumeCtx& ctx,
                       HitInfo* out) const {
        Node chunk;
        if (!get_chunk({(uint32_t)block_coordinates.x, (uint32_t)block_coordinates.y, (uint32_t)block_coordinates.z}, &chunk))
            return false;  // expect("out of range")
        P3i block{floor_size(block_coordinates.x, chunk.size), floor_size(block_coordinates.y, chunk.size),
                  floor_size(block_coordinates.z, chunk.size)};
        double cur[3] = {ray.origin[0], ray.origin[1], ray.origin[2]};
        const double* d = ray.direction;
        VolumeState vs;
        // the starting cell: a volume starts the free-path walk, a solid is ignored (ray_trace.rs:476-510)
        if (chunk.kind == Kind::Leaf && ctx.voxels && ctx.voxels[chunk.leaf].kind == MRT_VOX_VOLUME) {
            if (handle_volume(chunk.leaf, vs, cur, d, initial_normal, ctx, out) != Step::Continue) return true;
        }
        double sign_vec[3] = {std::signbit(d[0]) ? 0.0 : 1.0, std::signbit(d[1]) ? 0.0 : 1.0, std::signbit(d[2]) ? 0.0 : 1.0};
        for (int it = 0; it < 3000; ++it) {
            ++oct_step_counter();
            int step_size = get_step_size(block);
            if (step_size < 0) return false;  // get_step_size would panic; in_range is checked right after
            if (!in_range(block)) return false;
            int32_t bc[3] = {block.x, block.y, block.z};
            double t[3];
            for (int a = 0; a < 3; ++a) t[a] = ((double)bc[a] + (double)step_size * sign_vec[a] - cur[a]) / d[a];
            if (t[0] < 0.0 || t[1] < 0.0 || t[2] < 0.0) return false;  // logs an error, OutOfRange
            int axis;
            if (t[0] < t[1] && t[0] < t[2]) axis = 0;
            else if (t[1] < t[0] && t[1] < t[2]) axis = 1;
            else axis = 2;
            const int o1 = axis == 0 ? 1 : 0, o2 = axis == 2 ? 1 : 2;
            // the two other coordinates advance first
            cur[o1] += t[axis] * d[o1];
            cur[o2] += t[axis] * d[o2];
            double normal[3] = {0.0, 0.0, 0.0};
            Node node;
            if (d[axis] >= 0.0) {
                bc[axis] += step_size * int_sign(d[axis]);
                cur[axis] += t[axis] * d[axis];
                P3i probe;
                int32_t pc[3];
                pc[axis] = bc[axis];
                pc[o1] = f64_as_i32(cur[o1]);
                pc[o2] = f64_as_i32(cur[o2]);
                probe = {pc[0], pc[1], pc[2]};
                if (!in_range(probe)) return false;
                int ncs = get_chunk_size(probe);
                bc[o1] = floor_size(f64_as_i32(cur[o1]), (uint32_t)ncs);
                bc[o2] = floor_size(f64_as_i32(cur[o2]), (uint32_t)ncs);
                if (!get_chunk({(uint32_t)bc[0], (uint32_t)bc[1], (uint32_t)bc[2]}, &node)) return false;  // expect
                normal[axis] = -1.0;
            } else {
                int32_t pc[3] = {bc[0], bc[1], bc[2]};
                pc[axis] -= 1;
                if (!in_range({pc[0], pc[1], pc[2]})) return false;
                cur[axis] += t[axis] * d[axis];
                int32_t qc[3];
                qc[axis] = bc[axis] - 1;
                qc[o1] = f64_as_i32(cur[o1]);
                qc[o2] = f64_as_i32(cur[o2]);
                int ncs = get_chunk_size({qc[0], qc[1], qc[2]});
                if (ncs < 0) return false;  // panic in the reference
                int nss = 1 << ncs;
                bc[o1] = floor_size(f64_as_i32(cur[o1]), (uint32_t)ncs);
                bc[o2] = floor_size(f64_as_i32(cur[o2]), (uint32_t)ncs);
                bc[axis] -= nss;
                if (!get_chunk({(uint32_t)bc[0], (uint32_t)bc[1], (uint32_t)bc[2]}, &node)) return false;  // expect
                normal[axis] = 1.0;
            }
            block = {bc[0], bc[1], bc[2]};
            // handle_hit, ray_trace.rs:202-221
            Step s = Step::Continue;
            if (node.kind == Kind::Leaf) {
                if (ctx.voxels && ctx.voxels[node.leaf].kind == MRT_VOX_VOLUME) {
                    s = handle_volume(node.leaf, vs, cur, d, normal, ctx, out);
                } else {  // StopIterationSolid with the voxel's own material
                    out->solid = true, out->edge = false;
                    out->voxel = node.leaf;
                    for (int a = 0; a < 3; ++a) out->hit_position[a] = cur[a], out->normal[a] = normal[a];
                    return true;
                }
            } else {
                s = handle_empty(vs, cur, d, out);
            }
            if (s != Step::Continue) return true;
        }
        return false;  // "Max number of iterations reached"
    }
    bool ray_iteration(P3i block_coordinates, const Ray& ray, HitInfo* out) const {
        static const double unit_x[3] = {1.0, 0.0, 0.0};
        return ray_iteration(block_coordinates, ray, unit_x, VolumeCtx{}, out);
    }

    // trace_ray, ray_trace.rs:561-683. Returns false for None.
    bool trace_ray(const Ray& ray, HitInfo* out) const { return trace_ray(ray, VolumeCtx{}, out); }
    bool trace_ray(const Ray& ray, const VolumeCtx& ctx, HitInfo* out) const {
        static const double unit_x[3] = {1.0, 0.0, 0.0};
        static const int32_t BLOCK_OFFSETS[6][3] = {{0, 0, 0}, {-1, 0, 0}, {0, 0, 0}, {0, -1, 0}, {0, 0, 0}, {0, 0, -1}};
        static const double NORMALS[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
        double ws = (double)world_size();
        const double* o = ray.origin;
        if (o[0] >= 0.0 && o[0] <= ws && o[1] >= 0.0 && o[1] <= ws && o[2] >= 0.0 && o[2] <= ws) {
            return ray_iteration({f64_as_i32(o[0]), f64_as_i32(o[1]), f64_as_i32(o[2])}, ray, unit_x, ctx, out);
        }
        double values[6];
        for (int a = 0; a < 3; ++a) {
            values[2 * a] = (0.0 - o[a]) / ray.direction[a];
            values[2 * a + 1] = (ws - o[a]) / ray.direction[a];
        }
        int min_x = values[0] < values[1] ? 0 : 1, max_x = 1 - min_x;
        int min_y = values[2] < values[3] ? 2 : 3, max_y = 5 - min_y;
        int min_z = values[4] < values[5] ? 4 : 5, max_z = 9 - min_z;
        int min_index, max_index;
        if (values[min_x] > values[min_y] && values[min_x] > values[min_z]) min_index = min_x;
        else if (values[min_y] > values[min_x] && values[min_y] > values[min_z]) min_index = min_y;
        else min_index = min_z;
        if (values[max_x] < values[max_y] && values[max_x] < values[max_z]) max_index = max_x;
        else if (values[max_y] < values[max_x] && values[max_y] < values[max_z]) max_index = max_y;
        else max_index = max_z;
        if (!(values[max_index] >= values[min_index] && values[min_index] >= 0.0)) return false;
        double position[3];
        for (int a = 0; a < 3; ++a) position[a] = o[a] + values[min_index] * ray.direction[a];
        P3i bc{f64_as_i32(position[0]) + BLOCK_OFFSETS[min_index][0], f64_as_i32(position[1]) + BLOCK_OFFSETS[min_index][1],
               f64_as_i32(position[2]) + BLOCK_OFFSETS[min_index][2]};
        int32_t v;
        // `block_coordinate.map(|v| v as u32)`: a negative coordinate wraps to a huge value -> get() = None
        // a solid entry voxel is the hit; a volume or an empty cell starts the walk (ray_trace.rs:651-678)
        if (get({(uint32_t)bc.x, (uint32_t)bc.y, (uint32_t)bc.z}, &v) && !(ctx.voxels && ctx.voxels[v].kind == MRT_VOX_VOLUME)) {
            out->solid = true, out->edge = false;
            out->voxel = v;
            for (int a = 0; a < 3; ++a) out->hit_position[a] = position[a], out->normal[a] = NORMALS[min_index][a];
            return true;
        }
        Ray r2 = ray;
        for (int a = 0; a < 3; ++a) r2.origin[a] = position[a];
        r2.time = 0.0;
        return ray_iteration(bc, r2, NORMALS[min_index], ctx, out);
    }

    // Linearised image (one record per reachable node), for handing the tree to the ABI.
    void linearise(std::vector<mrt_oct_node>* nodes, std::vector<uint32_t>* children, uint32_t* root) const {
        nodes->clear();
        children->clear();
        *root = 0;
        const Node* r = arena.get_root();
        if (!r) return;
        std::vector<const Node*> order{r};
        for (size_t i = 0; i < order.size(); ++i) {
            const Node* n = order[i];
            mrt_oct_node rec{};
            rec.size = n->size;
            rec.first_child = -1;
            rec.voxel = -1;
            if (n->kind == Kind::Parent) {
                rec.kind = MRT_OCT_PARENT;
                rec.first_child = (int32_t)children->size();
                for (int c = 0; c < 8; ++c) {
                    children->push_back((uint32_t)order.size());
                    order.push_back(&arena.get_unchecked(n->children[c]));
                }
            } else if (n->kind == Kind::Leaf) {
                rec.kind = MRT_OCT_LEAF;
                rec.voxel = n->leaf;
            } else {
                rec.kind = MRT_OCT_EMPTY;
            }
            nodes->push_back(rec);
        }
    }

    // Builds from the ABI's linearised form (used when a scene descriptor carries an octree).
    bool from_linear(const mrt_octree& o, std::string* err) {
        arena = Arena{};
        if (o.n_nodes == 0) return true;
        if (o.root >= o.n_nodes) {
            *err = "octree: bad root";
            return false;
        }
        // arena slot 0 must be the root: emit breadth-first from the root
        std::vector<uint32_t> order{o.root};
        std::vector<size_t> slot(o.n_nodes, (size_t)-1);
        slot[o.root] = 0;
        for (size_t i = 0; i < order.size(); ++i) {
            const mrt_oct_node& n = o.nodes[order[i]];
            if (n.kind == MRT_OCT_PARENT) {
                if (n.first_child < 0 || (uint32_t)n.first_child + 8 > o.n_children) {
                    *err = "octree: bad child block";
                    return false;
                }
                for (int c = 0; c < 8; ++c) {
                    uint32_t ci = o.children[(uint32_t)n.first_child + (uint32_t)c];
                    if (ci >= o.n_nodes || slot[ci] != (size_t)-1) {
                        *err = "octree: child index out of range or shared";
                        return false;
                    }
                    slot[ci] = order.size();
                    order.push_back(ci);
                }
            }
        }
        arena.data.resize(order.size());
        for (size_t i = 0; i < order.size(); ++i) {
            const mrt_oct_node& n = o.nodes[order[i]];
            Node nd;
            nd.size = n.size;
            if (n.kind == MRT_OCT_PARENT) {
                nd.kind = Kind::Parent;
                for (int c = 0; c < 8; ++c) nd.children[c] = {slot[o.children[(uint32_t)n.first_child + (uint32_t)c]], 0};
            } else if (n.kind == MRT_OCT_LEAF) {
                nd.kind = Kind::Leaf;
                nd.leaf = n.voxel;
            } else {
                nd.kind = Kind::Empty;
            }
            arena.data[i] = {nd, 0};
        }
        return true;
    }
};

// prelude/rgb_color.rs:92-126
inline bool from_color_hex(const char* s, float rgb[3]) {
    if (!s || s[0] != '#') return false;  // expect("invalid syntax")
    size_t len = std::strlen(s + 1);
    if (len < 6) return false;  // assert!
    uint32_t total = 0;
    for (int i = 0; i < 6; ++i) {
        char c = s[1 + i];
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
        uint32_t number;
        if (c >= '0' && c <= '9') number = (uint32_t)(c - '0');
        else if (c >= 'a' && c <= 'f') number = (uint32_t)(c - 'a' + 10);
        else return false;  // panic!("invalid char")
        uint32_t p = 1;
        for (int k = 0; k < 5 - i; ++k) p *= 16u;
        total += number * p;
    }
    rgb[0] = (float)((total & 0xff0000) >> 16) / 255.f;
    rgb[1] = (float)((total & 0x00ff00) >> 8) / 255.f;
    rgb[2] = (float)(total & 0x0000ff) / 255.f;
    return true;
}

}  // namespace octref
#endif
