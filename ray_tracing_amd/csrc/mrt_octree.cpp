// mrt_octree.cpp -- host-side construction of the linearised voxel octree (the step before the path
// for voxel scenes). The reference builds FastOctTree<Voxel> by repeated `set` calls that split and
// re-collapse nodes along one root-to-leaf path per voxel with node clones on the way
// (fast_oct_tree/mod.rs:43-169, operations/set.rs:4-36); the tree it ends with is canonical: a node
// whose eight children are equal leaves (or all empty) is that leaf (or empty), and the world edge is
// the smallest power of two above the largest coordinate ever set. This builder keeps the voxel set
// in a hash map (last write wins) and produces that canonical tree bottom-up from the Morton-sorted
// set in O(n log n), then emits it in the layout of mrt_octree.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "mrt_internal.h"

namespace {

inline uint64_t pack(uint32_t x, uint32_t y, uint32_t z) { return ((uint64_t)x << 42) | ((uint64_t)y << 21) | (uint64_t)z; }
constexpr uint32_t COORD_MAX = (1u << 21) - 1;

struct VoxelKey {  // exact-value identity of a voxel record
    unsigned char bytes[sizeof(mrt_voxel)];
    bool operator==(const VoxelKey& o) const { return std::memcmp(bytes, o.bytes, sizeof(bytes)) == 0; }
};
struct VoxelKeyHash {
    size_t operator()(const VoxelKey& k) const {
        uint64_t h = 1469598103934665603ull;
        for (unsigned char b : k.bytes) h = (h ^ b) * 1099511628211ull;
        return (size_t)h;
    }
};

}  // namespace

struct mrt_voxel_grid {
    std::unordered_map<uint64_t, uint32_t> cells;  // packed position -> voxel record index
    std::vector<mrt_voxel> table;
    std::unordered_map<VoxelKey, uint32_t, VoxelKeyHash> table_index;
    int64_t max_coord = -1;  // largest coordinate ever set (also by voxels that were overwritten later)
    // linearised image (valid until the next mutation)
    std::vector<mrt_oct_node> nodes;
    std::vector<uint32_t> children;
    std::vector<mrt_voxel> used_voxels;

    uint32_t intern(const mrt_voxel& v) {
        mrt_voxel clean = v;
        clean.reserved = 0;
        if (clean.kind != MRT_VOX_VOLUME) clean.edge_kind = 0, clean.edge_probability = 0, clean.edge_fuzz = 0,
                                          clean.edge_albedo[0] = clean.edge_albedo[1] = clean.edge_albedo[2] = 0, clean.density = 0;
        VoxelKey k;
        std::memcpy(k.bytes, &clean, sizeof(clean));
        auto it = table_index.find(k);
        if (it != table_index.end()) return it->second;
        uint32_t id = (uint32_t)table.size();
        table.push_back(clean);
        table_index.emplace(k, id);
        return id;
    }
    void set(uint32_t x, uint32_t y, uint32_t z, uint32_t voxel) {
        cells[pack(x, y, z)] = voxel;
        max_coord = std::max<int64_t>(max_coord, std::max(x, std::max(y, z)));
    }
};

namespace {

int fail(int code, const char* msg) {
    mrt::set_last_error(msg);
    return code;
}

// Minimal MagicaVoxel reader with dot_vox 5.1.1's conventions for what load_vox uses: every XYZI chunk
// is a model, a voxel's `i` is the file's colour index minus one, the palette is the RGBA chunk.
struct VoxFile {
    std::vector<std::vector<uint8_t>> models;  // x y z i quadruples
    std::vector<uint8_t> palette;              // 256 * 4
};
bool parse_vox(const uint8_t* data, size_t len, VoxFile* out, const char** why) {
    auto rd32 = [&](size_t p) { return (uint32_t)data[p] | ((uint32_t)data[p + 1] << 8) | ((uint32_t)data[p + 2] << 16) | ((uint32_t)data[p + 3] << 24); };
    if (len < 20 || std::memcmp(data, "VOX ", 4) != 0) return *why = "not a .vox file", false;
    if (std::memcmp(data + 8, "MAIN", 4) != 0) return *why = "missing MAIN chunk", false;
    size_t n = rd32(12), m = rd32(16);
    size_t p = 20 + n, end = std::min(len, 20 + n + m);
    while (p + 12 <= end) {
        const uint8_t* id = data + p;
        size_t cn = rd32(p + 4), cm = rd32(p + 8), body = p + 12;
        if (body + cn > len) return *why = "truncated chunk", false;
        if (std::memcmp(id, "XYZI", 4) == 0) {
            if (cn < 4) return *why = "bad XYZI chunk", false;
            size_t cnt = rd32(body);
            if (4 + cnt * 4 > cn) return *why = "bad XYZI chunk", false;
            std::vector<uint8_t> v(data + body + 4, data + body + 4 + cnt * 4);
            for (size_t i = 0; i < cnt; ++i) v[4 * i + 3] = (uint8_t)(v[4 * i + 3] - 1);
            out->models.push_back(std::move(v));
        } else if (std::memcmp(id, "RGBA", 4) == 0) {
            if (cn < 1024) return *why = "bad RGBA chunk", false;
            out->palette.assign(data + body, data + body + 1024);
        }
        p = body + cn + cm;
    }
    if (out->palette.empty()) return *why = ".vox without an RGBA chunk (default palette) is not supported", false;
    return true;
}

}  // namespace

extern "C" {

mrt_voxel_grid* mrt_voxels_create(void) { return new (std::nothrow) mrt_voxel_grid(); }
void mrt_voxels_destroy(mrt_voxel_grid* g) { delete g; }

int mrt_voxels_set(mrt_voxel_grid* g, const int32_t* xyz, const mrt_voxel* voxels, size_t n, int one_voxel_for_all) {
    if (!g || (n && (!xyz || !voxels))) return fail(MRT_ERR_INVALID_ARGUMENT, "NULL argument");
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a)
            if (xyz[3 * i + a] < 0 || (uint32_t)xyz[3 * i + a] > COORD_MAX)
                return fail(MRT_ERR_INVALID_ARGUMENT, "voxel coordinate out of range (0 .. 2^21 - 1)");
    try {
        uint32_t shared = one_voxel_for_all && n ? g->intern(voxels[0]) : 0;
        for (size_t i = 0; i < n; ++i) {
            const mrt_voxel& v = voxels[one_voxel_for_all ? 0 : i];
            if (v.kind < MRT_VOX_LAMBERTIAN || v.kind > MRT_VOX_VOLUME) return fail(MRT_ERR_INVALID_ARGUMENT, "unknown voxel kind");
            g->set((uint32_t)xyz[3 * i], (uint32_t)xyz[3 * i + 1], (uint32_t)xyz[3 * i + 2], one_voxel_for_all ? shared : g->intern(v));
        }
    } catch (const std::bad_alloc&) {
        return fail(MRT_ERR_OUT_OF_MEMORY, "out of host memory");
    }
    return MRT_OK;
}

int mrt_voxels_combine(mrt_voxel_grid* dst, const mrt_voxel_grid* src, const int32_t offset[3]) {
    if (!dst || !src || !offset || dst == src) return fail(MRT_ERR_INVALID_ARGUMENT, "bad argument");
    try {
        // combine.rs:4-19 sets other's voxels in x, y, z order; the order only matters between voxels of
        // `other`, which never collide with each other
        for (const auto& kv : src->cells) {
            int64_t x = (int64_t)(kv.first >> 42) + offset[0], y = (int64_t)((kv.first >> 21) & COORD_MAX) + offset[1],
                    z = (int64_t)(kv.first & COORD_MAX) + offset[2];
            if (x < 0 || y < 0 || z < 0) continue;  // positions that become negative are dropped
            if (x > COORD_MAX || y > COORD_MAX || z > COORD_MAX) return fail(MRT_ERR_INVALID_ARGUMENT, "voxel coordinate out of range");
            dst->set((uint32_t)x, (uint32_t)y, (uint32_t)z, dst->intern(src->table[kv.second]));
        }
    } catch (const std::bad_alloc&) {
        return fail(MRT_ERR_OUT_OF_MEMORY, "out of host memory");
    }
    return MRT_OK;
}

int mrt_voxels_load_vox(mrt_voxel_grid* g, const uint8_t* data, size_t len) {
    if (!g || !data) return fail(MRT_ERR_INVALID_ARGUMENT, "NULL argument");
    VoxFile vf;
    const char* why = "";
    if (!parse_vox(data, len, &vf, &why)) return fail(MRT_ERR_INVALID_ARGUMENT, why);
    // load.rs:74-127: all models overlaid, scene-graph transforms ignored
    int mn[3] = {256, 256, 256}, mx[3] = {-1, -1, -1};
    for (const auto& m : vf.models)
        for (size_t i = 0; i + 3 < m.size(); i += 4)
            for (int a = 0; a < 3; ++a) mn[a] = std::min(mn[a], (int)m[i + a]), mx[a] = std::max(mx[a], (int)m[i + a]);
    try {
        for (const auto& m : vf.models)
            for (size_t i = 0; i + 3 < m.size(); i += 4) {
                const uint8_t* c = &vf.palette[(size_t)m[i + 3] * 4];
                mrt_voxel v{};
                v.kind = MRT_VOX_LAMBERTIAN;
                v.albedo[0] = (float)c[0] / 255.0f, v.albedo[1] = (float)c[1] / 255.0f, v.albedo[2] = (float)c[2] / 255.0f;
                int x = (int)m[i] - mn[0], y = (int)m[i + 2] - mn[2], z = -((int)m[i + 1] - mn[1]) + mx[1] - mn[1];
                g->set((uint32_t)x, (uint32_t)y, (uint32_t)z, g->intern(v));
            }
    } catch (const std::bad_alloc&) {
        return fail(MRT_ERR_OUT_OF_MEMORY, "out of host memory");
    }
    return MRT_OK;
}

size_t mrt_voxels_count(const mrt_voxel_grid* g) { return g ? g->cells.size() : 0; }

int mrt_voxels_linearise(mrt_voxel_grid* g, mrt_octree* out) {
    if (!g || !out) return fail(MRT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (g->max_coord < 0) return fail(MRT_ERR_INVALID_ARGUMENT, "empty octree has no bounding box");
    try {
        uint32_t k = 0;
        while ((1ll << k) <= g->max_coord) ++k;  // world edge 2^k
        // compact the voxel table to the records that are still in use
        std::vector<int32_t> remap(g->table.size(), -1);
        g->used_voxels.clear();
        struct Cell {
            uint64_t key;      // packed position at the current level
            int32_t kind;      // MRT_OCT_LEAF / MRT_OCT_PARENT
            int64_t payload;   // leaf: voxel id; parent: index of its first child record
        };
        std::vector<Cell> cur;
        cur.reserve(g->cells.size());
        for (const auto& kv : g->cells) {
            if (remap[kv.second] < 0) {
                remap[kv.second] = (int32_t)g->used_voxels.size();
                g->used_voxels.push_back(g->table[kv.second]);
            }
            cur.push_back(Cell{kv.first, MRT_OCT_LEAF, remap[kv.second]});
        }
        // canonical order of the voxel table does not matter; order cells by position
        std::sort(cur.begin(), cur.end(), [](const Cell& a, const Cell& b) { return a.key < b.key; });
        g->nodes.clear();
        g->children.clear();
        auto parent_key = [](uint64_t key) {
            uint32_t x = (uint32_t)(key >> 42), y = (uint32_t)((key >> 21) & COORD_MAX), z = (uint32_t)(key & COORD_MAX);
            return pack(x >> 1, y >> 1, z >> 1);
        };
        auto slot_of = [](uint64_t key) {
            uint32_t x = (uint32_t)(key >> 42), y = (uint32_t)((key >> 21) & COORD_MAX), z = (uint32_t)(key & COORD_MAX);
            return ((x & 1u) << 2) | ((y & 1u) << 1) | (z & 1u);  // Node::pos_to_index, mod.rs: x*4 + y*2 + z
        };
        for (uint32_t level = 1; level <= k; ++level) {
            // cells of one parent are not contiguous in packed-position order: group by parent key
            std::vector<std::pair<uint64_t, uint32_t>> order(cur.size());
            for (uint32_t i = 0; i < cur.size(); ++i) order[i] = {parent_key(cur[i].key), i};
            std::sort(order.begin(), order.end());
            std::vector<Cell> next;
            for (size_t b = 0; b < order.size();) {
                size_t e = b;
                while (e < order.size() && order[e].first == order[b].first) ++e;
                int32_t ck[8];
                int64_t cp[8];
                for (int s = 0; s < 8; ++s) ck[s] = MRT_OCT_EMPTY, cp[s] = -1;
                for (size_t i = b; i < e; ++i) {
                    const Cell& c = cur[order[i].second];
                    uint32_t s = slot_of(c.key);
                    ck[s] = c.kind, cp[s] = c.payload;
                }
                bool collapse = true;
                for (int s = 0; s < 8; ++s) collapse = collapse && ck[s] == MRT_OCT_LEAF && cp[s] == cp[0];
                if (collapse) {
                    next.push_back(Cell{order[b].first, MRT_OCT_LEAF, cp[0]});
                } else {
                    int64_t first = (int64_t)g->children.size();
                    for (int s = 0; s < 8; ++s) {
                        mrt_oct_node rec{};
                        rec.kind = ck[s], rec.size = level - 1;
                        rec.first_child = ck[s] == MRT_OCT_PARENT ? (int32_t)cp[s] : -1;
                        rec.voxel = ck[s] == MRT_OCT_LEAF ? (int32_t)cp[s] : -1;
                        g->children.push_back((uint32_t)g->nodes.size());
                        g->nodes.push_back(rec);
                    }
                    next.push_back(Cell{order[b].first, MRT_OCT_PARENT, first});
                }
                b = e;
            }
            cur.swap(next);
        }
        // the root
        mrt_oct_node root{};
        root.kind = cur[0].kind, root.size = k;
        root.first_child = cur[0].kind == MRT_OCT_PARENT ? (int32_t)cur[0].payload : -1;
        root.voxel = cur[0].kind == MRT_OCT_LEAF ? (int32_t)cur[0].payload : -1;
        g->nodes.push_back(root);
        out->root = (uint32_t)g->nodes.size() - 1;
        out->n_nodes = (uint32_t)g->nodes.size();
        out->nodes = g->nodes.data();
        out->n_children = (uint32_t)g->children.size();
        out->children = g->children.data();
        out->n_voxels = (uint32_t)g->used_voxels.size();
        out->voxels = g->used_voxels.data();
    } catch (const std::bad_alloc&) {
        return fail(MRT_ERR_OUT_OF_MEMORY, "out of host memory");
    }
    return MRT_OK;
}

}  // extern "C"
