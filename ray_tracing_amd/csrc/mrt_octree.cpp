// mrt_octree.cpp -- host-side construction of the linearised voxel octree (the step before the path
// for voxel scenes). The reference builds FastOctTree<Voxel> by repeated `set` calls that split and
// re-collapse nodes along one root-to-leaf path per voxel with node clones on the way
// (fast_oct_tree/mod.rs:43-169, operations/set.rs:4-36); the tree it ends with is canonical -- a node
// whose eight children are equal leaves (or all empty) is that leaf (or empty), and the world edge is
// the smallest power of two above the largest coordinate ever set -- except for the chain of parents
// with eight empty children that growing the world leaves in the origin corner (see the end of
// mrt_voxels_linearise). This builder keeps the voxel set as a key-sorted array plus the sets made since (last write wins),
// produces the canonical tree bottom-up from the sorted set -- one sort, then one linear pass per level --
// adds that chain, and emits the result in the layout of mrt_octree.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "mrt_internal.h"

namespace {

constexpr uint32_t COORD_MAX = (1u << 21) - 1;

// Cell key: the three 21-bit coordinates interleaved so that, per bit position b, the reference's child
// index x*4 + y*2 + z (Node::pos_to_index, mod.rs:195-207) of bit b sits at bits 3b+2 .. 3b. Sorting by
// this key puts the cells of every tree node next to each other at every level: parent = key >> 3,
// child slot = key & 7.
inline uint64_t spread21(uint32_t v) {
    uint64_t x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
inline uint32_t compact21(uint64_t x) {
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10C30C30C30C30C3ull;
    x = (x ^ (x >> 4)) & 0x100F00F00F00F00Full;
    x = (x ^ (x >> 8)) & 0x1F0000FF0000FFull;
    x = (x ^ (x >> 16)) & 0x1F00000000FFFFull;
    x = (x ^ (x >> 32)) & 0x1FFFFFull;
    return (uint32_t)x;
}
inline uint64_t cell_key(uint32_t x, uint32_t y, uint32_t z) { return (spread21(x) << 2) | (spread21(y) << 1) | spread21(z); }
inline void cell_xyz(uint64_t key, uint32_t& x, uint32_t& y, uint32_t& z) { x = compact21(key >> 2), y = compact21(key >> 1), z = compact21(key); }

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

struct Cell {
    uint64_t key;
    uint32_t voxel;  // index into mrt_voxel_grid::table
};

}  // namespace

// The voxel set: `cells` is sorted by key with one entry per position; `pending` holds the sets made
// since it was last brought up to date, in call order (a later set of a position overwrites).
struct mrt_voxel_grid {
    std::vector<Cell> cells, pending;
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
        pending.push_back(Cell{cell_key(x, y, z), voxel});
        max_coord = std::max<int64_t>(max_coord, std::max(x, std::max(y, z)));
    }
    // Folds `pending` into `cells`: the last set of a position wins, also over what `cells` held.
    void settle() {
        if (pending.empty()) return;
        std::stable_sort(pending.begin(), pending.end(), [](const Cell& a, const Cell& b) { return a.key < b.key; });
        size_t n = 0;
        for (size_t i = 0; i < pending.size(); ++i)
            if (i + 1 == pending.size() || pending[i + 1].key != pending[i].key) pending[n++] = pending[i];
        pending.resize(n);
        if (cells.empty()) {
            cells.swap(pending);
        } else {
            std::vector<Cell> merged;
            merged.reserve(cells.size() + pending.size());
            size_t a = 0, b = 0;
            while (a < cells.size() || b < pending.size()) {
                if (b == pending.size() || (a < cells.size() && cells[a].key < pending[b].key)) merged.push_back(cells[a++]);
                else {
                    if (a < cells.size() && cells[a].key == pending[b].key) ++a;
                    merged.push_back(pending[b++]);
                }
            }
            cells.swap(merged);
        }
        pending.clear();
        pending.shrink_to_fit();
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
        std::vector<Cell> src_cells = src->cells, src_pending = src->pending;
        if (!src_pending.empty()) {  // bring a private copy of the source up to date
            mrt_voxel_grid tmp;
            tmp.cells.swap(src_cells), tmp.pending.swap(src_pending);
            tmp.settle();
            src_cells.swap(tmp.cells);
        }
        std::vector<int64_t> remap(src->table.size(), -1);
        for (const Cell& c : src_cells) {
            uint32_t sx, sy, sz;
            cell_xyz(c.key, sx, sy, sz);
            int64_t x = (int64_t)sx + offset[0], y = (int64_t)sy + offset[1], z = (int64_t)sz + offset[2];
            if (x < 0 || y < 0 || z < 0) continue;  // positions that become negative are dropped
            if (x > COORD_MAX || y > COORD_MAX || z > COORD_MAX) return fail(MRT_ERR_INVALID_ARGUMENT, "voxel coordinate out of range");
            if (remap[c.voxel] < 0) remap[c.voxel] = dst->intern(src->table[c.voxel]);
            dst->set((uint32_t)x, (uint32_t)y, (uint32_t)z, (uint32_t)remap[c.voxel]);
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

size_t mrt_voxels_count(const mrt_voxel_grid* g) {
    if (!g) return 0;
    try {
        const_cast<mrt_voxel_grid*>(g)->settle();  // representation only: the voxel set is unchanged
    } catch (const std::bad_alloc&) {
        return 0;
    }
    return g->cells.size();
}

int mrt_voxels_linearise(mrt_voxel_grid* g, mrt_octree* out) {
    if (!g || !out) return fail(MRT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (g->max_coord < 0) return fail(MRT_ERR_INVALID_ARGUMENT, "empty octree has no bounding box");
    try {
        uint32_t k = 0;
        while ((1ll << k) <= g->max_coord) ++k;  // world edge 2^k
        g->settle();
        // compact the voxel table to the records that are still in use
        std::vector<int32_t> remap(g->table.size(), -1);
        g->used_voxels.clear();
        struct Level {
            uint64_t key;      // cell key at the current level
            int32_t kind;      // MRT_OCT_LEAF / MRT_OCT_PARENT
            int64_t payload;   // leaf: voxel id; parent: index of its first child record
        };
        std::vector<Level> cur, next;
        cur.reserve(g->cells.size());
        for (const Cell& c : g->cells) {
            if (remap[c.voxel] < 0) {
                remap[c.voxel] = (int32_t)g->used_voxels.size();
                g->used_voxels.push_back(g->table[c.voxel]);
            }
            cur.push_back(Level{c.key, MRT_OCT_LEAF, remap[c.voxel]});
        }
        g->nodes.clear();
        g->children.clear();
        for (uint32_t level = 1; level <= k; ++level) {
            // `cur` is sorted by key, so the cells of one parent (key >> 3) are consecutive
            next.clear();
            for (size_t b = 0; b < cur.size();) {
                const uint64_t parent = cur[b].key >> 3;
                size_t e = b;
                int32_t ck[8];
                int64_t cp[8];
                for (int s = 0; s < 8; ++s) ck[s] = MRT_OCT_EMPTY, cp[s] = -1;
                for (; e < cur.size() && (cur[e].key >> 3) == parent; ++e) {
                    const uint32_t s = (uint32_t)(cur[e].key & 7u);  // Node::pos_to_index: x*4 + y*2 + z
                    ck[s] = cur[e].kind, cp[s] = cur[e].payload;
                }
                bool collapse = true;
                for (int s = 0; s < 8; ++s) collapse = collapse && ck[s] == MRT_OCT_LEAF && cp[s] == cp[0];
                if (collapse) {
                    next.push_back(Level{parent, MRT_OCT_LEAF, cp[0]});
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
                    next.push_back(Level{parent, MRT_OCT_PARENT, first});
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
        // The growth chain. `set` grows the world by wrapping the old root as child 0 of a new parent with
        // seven empty siblings (operations/set.rs:11-30), starting from an empty root of size 0, and
        // `set_child` re-collapses nodes only along the path of the voxel being set (mod.rs:129-165). So on the
        // path root -> child 0 -> child 0 ... every node that holds no voxel is, in the reference's tree, a
        // parent of eight empty children (down to size 1) where the canonical tree has one large empty node;
        // rays through the origin corner step through those small cells.
        for (uint32_t cur = out->root;;) {
            if (g->nodes[cur].kind == MRT_OCT_PARENT) {
                cur = g->children[(size_t)g->nodes[cur].first_child];
                continue;
            }
            if (g->nodes[cur].kind != MRT_OCT_EMPTY || g->nodes[cur].size < 1) break;
            const uint32_t child_size = g->nodes[cur].size - 1;
            g->nodes[cur].kind = MRT_OCT_PARENT;
            g->nodes[cur].first_child = (int32_t)g->children.size();
            const uint32_t first = (uint32_t)g->nodes.size();
            for (int s = 0; s < 8; ++s) {
                mrt_oct_node rec{};
                rec.kind = MRT_OCT_EMPTY, rec.size = child_size, rec.first_child = -1, rec.voxel = -1;
                g->children.push_back(first + (uint32_t)s);
                g->nodes.push_back(rec);
            }
            cur = first;
        }
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
