// octree_api.cpp -- TEST INFRASTRUCTURE ONLY: C entry points over octree_ref.h so that the
// reference's own unit-test vectors (arena.rs:118-239, mod.rs:265-342, operations/set.rs:38-173,
// operations/combine.rs:21-65, ray_trace.rs:699-855, hittable.rs:87-112, rgb_color.rs:238-297)
// can be replayed from pytest.
#include "octree_ref.h"

using namespace octref;

extern "C" {

void* octref_new() { return new Tree(); }
void octref_free(void* t) { delete static_cast<Tree*>(t); }
void octref_set(void* t, int32_t value, uint32_t x, uint32_t y, uint32_t z) { static_cast<Tree*>(t)->set(value, {x, y, z}); }
// returns 1 and writes *out for Some, 0 for None
int octref_get(void* t, uint32_t x, uint32_t y, uint32_t z, int32_t* out) { return static_cast<Tree*>(t)->get({x, y, z}, out) ? 1 : 0; }
uint32_t octref_world_size(void* t) { return static_cast<Tree*>(t)->world_size(); }
// root: kind (0 parent, 1 leaf, 2 empty), size, leaf payload; returns 0 when the arena is empty
int octref_root(void* t, int32_t* kind, uint32_t* size, int32_t* leaf) {
    const Node* r = static_cast<Tree*>(t)->arena.get_root();
    if (!r) return 0;
    *kind = (int32_t)r->kind, *size = r->size, *leaf = r->leaf;
    return 1;
}
int octref_get_chunk(void* t, uint32_t x, uint32_t y, uint32_t z, int32_t* kind, uint32_t* size, int32_t* leaf) {
    Node n;
    if (!static_cast<Tree*>(t)->get_chunk({x, y, z}, &n)) return 0;
    *kind = (int32_t)n.kind, *size = n.size, *leaf = n.leaf;
    return 1;
}
int octref_get_chunk_size(void* t, int32_t x, int32_t y, int32_t z) { return static_cast<Tree*>(t)->get_chunk_size({x, y, z}); }
int octref_get_step_size(void* t, int32_t x, int32_t y, int32_t z) { return static_cast<Tree*>(t)->get_step_size({x, y, z}); }
void octref_combine(void* dst, void* other, int32_t ox, int32_t oy, int32_t oz) {
    static_cast<Tree*>(dst)->combine(*static_cast<Tree*>(other), {ox, oy, oz});
}
void octref_arena_stats(void* t, uint64_t* data_len, uint64_t* num_deleted) {
    Tree* tr = static_cast<Tree*>(t);
    *data_len = tr->arena.data.size(), *num_deleted = tr->arena.deleted_indices.size();
}
void octref_set_many(void* t, const int32_t* xyzv, size_t n) {
    Tree* tr = static_cast<Tree*>(t);
    for (size_t i = 0; i < n; ++i)
        tr->set(xyzv[4 * i + 3], {(uint32_t)xyzv[4 * i], (uint32_t)xyzv[4 * i + 1], (uint32_t)xyzv[4 * i + 2]});
}
// trace_ray: ray = origin xyz, direction xyz (already normalised by the caller as hittable.rs:13-17 does)
int octref_trace_ray(void* t, const double* ray, double* hit_position, double* normal, int32_t* voxel) {
    Ray r{{ray[0], ray[1], ray[2]}, {ray[3], ray[4], ray[5]}, 0.0};
    HitInfo h;
    if (!static_cast<Tree*>(t)->trace_ray(r, &h)) return 0;
    for (int a = 0; a < 3; ++a) hit_position[a] = h.hit_position[a], normal[a] = h.normal[a];
    *voxel = h.voxel;
    return 1;
}
// linearise: two calls (query sizes with NULL buffers, then fill)
void octref_linearise(void* t, mrt_oct_node* nodes, size_t* n_nodes, uint32_t* children, size_t* n_children, uint32_t* root) {
    std::vector<mrt_oct_node> n;
    std::vector<uint32_t> c;
    static_cast<Tree*>(t)->linearise(&n, &c, root);
    if (nodes) std::memcpy(nodes, n.data(), n.size() * sizeof(mrt_oct_node));
    if (children) std::memcpy(children, c.data(), c.size() * sizeof(uint32_t));
    *n_nodes = n.size(), *n_children = c.size();
}

// ---- generational arena on its own (arena.rs tests use Arena<u8>/<u32>): payload in Node::leaf
void* octref_arena_new() { return new Arena(); }
void octref_arena_free(void* a) { delete static_cast<Arena*>(a); }
void octref_arena_insert(void* a, int32_t v, uint64_t* index, uint32_t* generation) {
    Node n;
    n.kind = Kind::Leaf;
    n.leaf = v;
    ArenaIndex i = static_cast<Arena*>(a)->insert(n);
    *index = i.index, *generation = i.generation;
}
int octref_arena_get(void* a, uint64_t index, uint32_t generation, int32_t* out) {
    const Node* n = static_cast<Arena*>(a)->get({(size_t)index, generation});
    if (!n) return 0;
    *out = n->leaf;
    return 1;
}
int octref_arena_key_exists(void* a, uint64_t index, uint32_t generation) {
    return static_cast<Arena*>(a)->key_exists({(size_t)index, generation}) ? 1 : 0;
}
int octref_arena_update(void* a, uint64_t index, uint32_t generation, int32_t v) {
    Node n;
    n.kind = Kind::Leaf;
    n.leaf = v;
    return static_cast<Arena*>(a)->update({(size_t)index, generation}, n) ? 1 : 0;
}
void octref_arena_update_root(void* a, int32_t v) {
    Node n;
    n.kind = Kind::Leaf;
    n.leaf = v;
    static_cast<Arena*>(a)->update_root(n);
}
int octref_arena_get_root(void* a, int32_t* out) {
    const Node* n = static_cast<Arena*>(a)->get_root();
    if (!n) return 0;
    *out = n->leaf;
    return 1;
}
int octref_arena_delete(void* a, uint64_t index, uint32_t generation) {
    return static_cast<Arena*>(a)->del({(size_t)index, generation}) ? 1 : 0;
}
void octref_arena_raw_stats(void* a, uint64_t* data_len, uint64_t* num_deleted) {
    *data_len = static_cast<Arena*>(a)->data.size(), *num_deleted = static_cast<Arena*>(a)->deleted_indices.size();
}

// ---- free functions
int32_t octref_int_sign(double a) { return int_sign(a); }
int32_t octref_floor_size(int32_t v, uint32_t s) { return floor_size(v, s); }
int octref_pos_to_index(uint32_t x, uint32_t y, uint32_t z) { return pos_to_index({x, y, z}); }
void octref_world_pos_to_child_pos(uint32_t x, uint32_t y, uint32_t z, uint32_t size, uint32_t* out) {
    P3u p = world_pos_to_child_pos({x, y, z}, size);
    out[0] = p.x, out[1] = p.y, out[2] = p.z;
}
int octref_world_pos_to_child_index(uint32_t x, uint32_t y, uint32_t z, uint32_t size) {
    return world_pos_to_child_index({x, y, z}, size);
}
int octref_from_color_hex(const char* s, float* rgb) { return from_color_hex(s, rgb) ? 1 : 0; }

}  // extern "C"
