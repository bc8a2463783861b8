// Test driver: runs the host-side scene flattening (reference-topology tree, node records, candidate
// tree with its ancestor table) on random many-object scenes under the address and undefined-
// behaviour sanitizers, and checks the structural invariants of the candidate tree.
// Built and run by tests/test_host_logic.py; not part of the library.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../ray_tracing_amd/csrc/mrt_internal.h"

namespace mrt {
void set_last_error(const std::string&) {}
}  // namespace mrt

static int fail(const char* what, int scene) {
    std::fprintf(stderr, "FAILED scene %d: %s\n", scene, what);
    return 1;
}

int main() {
    std::mt19937_64 rng(7);
    auto uni = [&](double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); };
    for (int scene = 0; scene < 40; ++scene) {
        const int n = scene < 4 ? scene + 1 : 2 + (int)(rng() % 520);
        std::vector<mrt_texture> tex(1);
        tex[0] = mrt_texture{};
        tex[0].kind = MRT_TEX_SOLID;
        std::vector<mrt_material> mat(1);
        mat[0] = mrt_material{};
        mat[0].kind = MRT_MAT_LAMBERTIAN;
        std::vector<mrt_shape> shapes;
        std::vector<mrt_object> objects;
        const bool with_medium = scene % 9 == 8;
        for (int i = 0; i < n; ++i) {
            mrt_shape s{};
            s.material = 0, s.boundary = -1, s.octree = -1;
            const int k = (int)(rng() % 6);
            const double c[3] = {uni(-12, 12), uni(0, 2), uni(-12, 12)};
            if (k <= 1) {
                s.kind = MRT_SHAPE_SPHERE;
                s.p[0] = (rng() % 17 == 0 ? -1.0 : 1.0) * uni(0.1, 0.6);
                s.p[1] = c[0], s.p[2] = c[1], s.p[3] = c[2];
            } else if (k <= 3) {
                s.kind = MRT_SHAPE_MOVING_SPHERE;
                s.p[0] = uni(0.1, 0.6);
                for (int a = 0; a < 3; ++a) s.p[1 + a] = c[a], s.p[4 + a] = c[a] + uni(-0.5, 0.5);
                s.p[7] = uni(-1, 0.5), s.p[8] = s.p[7] + uni(0.2, 2.0);
            } else if (k == 4) {
                s.kind = MRT_SHAPE_RENDER_BOX;
                for (int a = 0; a < 3; ++a) s.p[a] = c[a] - 0.3, s.p[3 + a] = c[a] + uni(0.1, 0.5);
            } else {
                s.kind = MRT_SHAPE_XZ_RECT;
                s.p[0] = c[0] - 0.5, s.p[1] = c[0] + 0.5, s.p[2] = c[2] - 0.5, s.p[3] = c[2] + 0.5, s.p[4] = c[1], s.p[5] = 1.0;
            }
            shapes.push_back(s);
            if (with_medium && i == n / 2 && (s.kind == MRT_SHAPE_SPHERE || s.kind == MRT_SHAPE_RENDER_BOX)) {
                mrt_shape m{};
                m.kind = MRT_SHAPE_CONSTANT_MEDIUM, m.material = 0, m.boundary = (int32_t)shapes.size() - 1, m.octree = -1;
                m.p[0] = 0.3;
                shapes.push_back(m);
            }
            mrt_object o{};
            o.shape = (int32_t)shapes.size() - 1;
            const double ang = rng() % 3 == 0 ? uni(-1, 1) : 0.0, cs = std::cos(ang), sn = std::sin(ang);
            const double m16[16] = {cs, 0, -sn, 0, 0, 1, 0, 0, sn, 0, cs, 0, uni(-1, 1), 0, uni(-1, 1), 1};  // column-major
            for (int a = 0; a < 16; ++a) o.world_transform[a] = m16[a];
            objects.push_back(o);
        }
        mrt_scene_desc d{};
        d.abi_version = MRT_ABI_VERSION;
        d.n_textures = 1, d.textures = tex.data();
        d.n_materials = 1, d.materials = mat.data();
        d.n_shapes = (uint32_t)shapes.size(), d.shapes = shapes.data();
        d.n_objects = (uint32_t)objects.size(), d.objects = objects.data();
        d.camera = mrt_camera_info{1.0, 40.0, {16, 4, 5}, {0, 0.5, 0}, {0, 1, 0}, 0.01, 14.0, 0.0, 1.0};
        d.background.kind = MRT_BG_SKY, d.background.intensity = 1.0;
        d.bvh_seed = (uint64_t)scene;
        mrt::FlatScene f;
        std::string err;
        if (mrt::flatten_scene(&d, &f, &err) != MRT_OK) return fail(err.c_str(), scene);

        // reference-topology tree: pre-order, skip links inside the array
        const int nn = (int)f.nodes.size();
        for (int i = 0; i < nn; ++i)
            if (f.nodes[i].skip <= i || f.nodes[i].skip > nn) return fail("skip link", scene);
        bool stochastic = false;
        for (const auto& s : shapes) stochastic = stochastic || s.kind == MRT_SHAPE_CONSTANT_MEDIUM;
        if (stochastic || nn > mrt::CAND_MAX_NODES) {
            if (!f.cand_nodes.empty()) return fail("candidate tree built for a scene that does not qualify", scene);
            continue;
        }
        if (f.cand_nodes.empty()) return fail("no candidate tree", scene);
        // every leaf of the tree appears exactly once among the candidate tree's leaves, and every node's
        // parent link names the inner node whose subtree holds it
        std::vector<int> seen((size_t)nn, 0);
        for (const mrt::DCand4& c : f.cand_nodes) {
            for (int k = 0; k < 4; ++k) {
                const int ch = c.child[k];
                if (ch >= 0 && ch >= (int)f.cand_nodes.size()) return fail("child index", scene);
                if (ch <= -2) {
                    const int leaf = -2 - ch;
                    if (leaf >= nn || f.nodes[leaf].object < 0 || seen[(size_t)leaf]++) return fail("leaf index", scene);
                    for (int a = 0; a < 3; ++a)
                        if (!(c.lo[a][k] <= c.hi[a][k])) return fail("leaf box", scene);
                }
            }
        }
        for (int i = 0; i < nn; ++i) {
            const mrt::DNode& nd = f.nodes[i];
            if (nd.object >= 0 && !seen[(size_t)i]) return fail("leaf missing from the candidate tree", scene);
            if (i == 0) {
                if (nd.parent != -1) return fail("root has a parent", scene);
                continue;
            }
            if (nd.parent < 0 || nd.parent >= i) return fail("parent link", scene);
            const mrt::DNode& p = f.nodes[nd.parent];
            if (p.object >= 0 || !(i < p.skip)) return fail("parent does not enclose the node", scene);
            for (int j = nd.parent + 1; j < i; ++j)   // no closer enclosing inner node
                if (f.nodes[j].object < 0 && i < f.nodes[j].skip) return fail("parent is not the innermost", scene);
            for (int a = 0; a < 3; ++a)               // inner boxes are nested
                if (nd.object < 0 && !(p.bmin[a] <= nd.bmin[a] && nd.bmax[a] <= p.bmax[a])) return fail("boxes not nested", scene);
        }
    }
    std::puts("host flattening: 40 scenes ok");

    // the native octree builder: random sets with overwrites, combine with negative offsets, linearise
    for (int round = 0; round < 12; ++round) {
        mrt_voxel_grid* a = mrt_voxels_create();
        mrt_voxel_grid* b = mrt_voxels_create();
        const int extent = 1 << (1 + round % 7);
        std::vector<int32_t> xyz;
        std::vector<mrt_voxel> vox;
        const int count = 1 + (int)(rng() % 4000);
        for (int i = 0; i < count; ++i) {
            for (int k = 0; k < 3; ++k) xyz.push_back((int32_t)(rng() % (uint64_t)extent));
            mrt_voxel v{};
            v.kind = MRT_VOX_LAMBERTIAN;
            v.albedo[0] = (float)(rng() % 3) * 0.25f, v.albedo[1] = 0.5f, v.albedo[2] = 0.5f;
            vox.push_back(v);
        }
        if (mrt_voxels_set(a, xyz.data(), vox.data(), (size_t)count, 0) != MRT_OK) return fail("voxels_set", round);
        if (mrt_voxels_set(b, xyz.data(), vox.data(), (size_t)(count / 2 + 1), 1) != MRT_OK) return fail("voxels_set one", round);
        const int32_t off[3] = {(int32_t)(rng() % 9) - 4, (int32_t)(rng() % 9) - 4, (int32_t)(rng() % 40)};
        if (mrt_voxels_combine(a, b, off) != MRT_OK) return fail("voxels_combine", round);
        mrt_octree oc{};
        if (mrt_voxels_linearise(a, &oc) != MRT_OK) return fail("voxels_linearise", round);
        if (oc.n_nodes == 0 || oc.root >= oc.n_nodes || oc.n_children % 8 != 0) return fail("octree header", round);
        size_t leaves_volume = 0;
        for (uint32_t i = 0; i < oc.n_nodes; ++i) {
            const mrt_oct_node& nd = oc.nodes[i];
            if (nd.kind == MRT_OCT_PARENT) {
                if (nd.first_child < 0 || (uint32_t)nd.first_child + 8 > oc.n_children) return fail("child block", round);
                for (int k = 0; k < 8; ++k)
                    if (oc.children[nd.first_child + k] >= oc.n_nodes) return fail("child index", round);
            } else if (nd.kind == MRT_OCT_LEAF) {
                if (nd.voxel < 0 || (uint32_t)nd.voxel >= oc.n_voxels) return fail("voxel index", round);
                leaves_volume += (size_t)1 << (3 * nd.size);
            }
        }
        if (leaves_volume != mrt_voxels_count(a)) return fail("leaf volume differs from the voxel count", round);
        mrt_voxels_destroy(a);
        mrt_voxels_destroy(b);
    }
    std::puts("octree builder: 12 rounds ok");
    return 0;
}
