// mrt_internal.h -- flattened scene records shared by the host-side scene
// builder (mrt_host.cpp) and the HIP kernels (mrt_device.hip).
#ifndef MRT_INTERNAL_H
#define MRT_INTERNAL_H

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mrt.h"

namespace mrt {

// Affine map p' = (a[0]*x + a[1]*y + a[2]*z) + a[3] per output row.
struct Affine {
    double r[3][4];
};

// Object = shape + Transform (cpu_rt/src/ray_tracer/hittable.rs:184-201) with
// the matrices Object::hit recomputes per hit (hittable.rs:215-220) precomputed.
struct DObject {
    int32_t shape;
    int32_t identity;    // 1: world_transform is the identity matrix
    Affine to_object;    // world_transform
    Affine to_world;     // world_transform^-1
    double normal_m[3][3];  // (upper-left 3x3 of world_transform)^-1, rows
};

struct DShape {
    int32_t kind;
    int32_t material;
    int32_t boundary;
    int32_t octree;
    double p[10];  // parameter slots as in mrt_shape; media also carry p[9] = -1/density
};

struct DMaterial {
    int32_t kind;
    int32_t texture;
    double fuzz;
    double index_refraction;
    float color[3];
    int32_t pad;
};

struct DTexture {
    int32_t kind, a, b, image;
    float color[3];
    int32_t pad;
};

struct DImage {
    int32_t width, height;
    uint64_t offset;  // float offset into the image pool
};

// Device octree node (8 bytes): nodes of one tree are stored breadth-first so that the eight
// children of a parent are consecutive; `link` is the index of the first child (PARENT) or of the
// voxel record (LEAF), both absolute indices into the shared pools.
// In bits 16..31 of `kind_size` a parent carries the kinds of its eight children, two bits each (child c in bits
// 16 + 2c: MRT_OCT_EMPTY, MRT_OCT_LEAF, MRT_OCT_PARENT, or 3 for a leaf whose voxel is a volume), so that a
// descent reads a child only when it is itself a parent; a leaf carries its own code (1, or 3 for a volume voxel)
// in bits 16..17.
struct DOctNode {
    uint32_t kind_size;  // kind | size << 8 | child kinds << 16
    int32_t link;
};
struct DOctree {
    uint32_t root;         // absolute index of the root node
    uint32_t world_log2;   // root size (log2 of the world edge)
    uint32_t n_nodes;      // 0: empty tree (no bounding box)
    uint32_t cached;       // 1: the root is a parent and world_log2 < 16, so a walk can keep its path of parents (16 levels) per lane
};

// Camera::new output (camera.rs:44-76)
struct DCamera {
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
    double lens_radius, start_time, end_time;
};

// BVH node flags
enum : int32_t {
    NODE_DYNAMIC_BOX = 1,  // leaf whose box depends on (t_min, t_max): MovingSphere (bvh.rs:142-145,180)
    NODE_TWICE = 2,        // span-1 leaf pair folded into one node; evaluate the object twice (bvh.rs:224-232)
    // (DNodeRec only) leaf whose shape test implies its box test: an untransformed rectangle or sphere. A
    // rectangle's box is the rectangle itself, 0.002 thick (rect.rs:88-93,208-213,333-338), a sphere's is centre
    // +- radius (sphere.rs:57-62); a hit has its parameter inside the query range and its point on the shape, so
    // every slab interval contains it. The wave walk goes straight to the shape. (Within rounding of a rectangle's
    // edge, or of the six points where a sphere touches its box, the two tests are computed from different roundings
    // of the same quantity -- there the answer depends on rounding with or without the box. A hit at t = NaN is no
    // such hit and still takes the box test.)
    NODE_BOX_IMPLIED = 4,
    // (DNodeRec only) what the leaf test branches on, next to the flags so that it arrives with the links: the
    // object transform is the identity / a pure translation (linear part exactly the identity), and in bits 8..15
    // the kind of a medium's boundary shape
    NODE_IDENTITY = 8,
    NODE_TRANSLATION = 16,
    NODE_KIND2_SHIFT = 8
};

// Candidate search: largest tree (nodes) it handles -- one bit per node and lane in LDS -- and the
// depth of the per-lane node stack of its spatial tree, also in LDS.
constexpr int CAND_WORDS = 32;
constexpr int CAND_MAX_NODES = CAND_WORDS * 32;
constexpr int CAND_STACK = 32;
constexpr float CAND_MIN_BOUND = 1e-6f;

// Device BVH node of the per-lane walk: one 64-byte line per node.
struct DNode {
    double bmin[3], bmax[3];
    int32_t object;  // >= 0 leaf, -1 inner
    int32_t skip;    // next node in pre-order when this subtree is left
    int32_t flags;   // NODE_*
    int32_t parent;  // index of the enclosing inner node, -1 for the root
};
static_assert(sizeof(DNode) == 64, "DNode must be one 64-byte line");

// What the per-lane walk needs of a leaf's object besides the box, stored per NODE (same index as DNode),
// so that a node visit requests both records at once instead of node -> object one after the other.
struct DLeafRec {
    int32_t kind, gkind, identity, octree;  // as in DObjRec (zero for inner nodes); identity: 1 identity, 2 pure translation
    double p[10];                           // geometry parameters (media: the boundary's, and p[9] = -1/density)
};
static_assert(sizeof(DLeafRec) == 96, "DLeafRec layout");

// Node of the spatial tree of the candidate search: four children with their boxes. The search is
// bound by the cache lines its per-lane (divergent) node loads touch, so a node is two 128-byte lines:
// the first holds the child links and the boxes at ray time 0, the second the velocity bounds, which only
// nodes with a moving child need. Bounds are f32, rounded OUTWARD from the f64 boxes on the host (lower
// bounds down, upper bounds up), so the stored boxes contain the exact ones and the search stays a
// superset filter; the device tests them in f32 with the rounding error added to the comparison.
// A child's box at ray time tau is [lo + min(tau * vlo, tau * vhi), hi + max(tau * vlo, tau * vhi)]: a
// MovingSphere's centre is linear in the ray's time (sphere.rs:139-142), and scattered rays carry arbitrary
// times (ray_tracer.rs:242,261 set time = hit t), so the bound has to hold for every tau.
struct alignas(128) DCand4 {
    int32_t child[4];  // >= 0: node index; -1: none; <= -2: leaf, index (-2 - child) in FlatScene::nodes
    float lo[3][4], hi[3][4];
    float bound[3];    // per axis: the largest |lo|, |hi| of the children (at least CAND_MIN_BOUND): scale of the device's f32 rounding bound
    float vel;         // the largest |velocity bound| of the children, any axis
    float vlo[3][4], vhi[3][4];  // velocity bounds per unit of ray time
    int32_t pad2[8];
};
static_assert(sizeof(DCand4) == 256, "DCand4 is two 128-byte lines");

// One self-contained record per device BVH node for the wave-synchronous walk: everything the
// box test and the leaf's intersection test need, so a node costs one batch of scalar loads
// instead of the chain node -> object -> shape (-> boundary shape).
struct DNodeRec {
    double bmin[3], bmax[3];
    int32_t object;    // >= 0 leaf, -1 inner
    int32_t skip;
    int32_t flags;     // NODE_*
    int32_t kind;      // shape kind of the leaf's object (CONSTANT_MEDIUM for media)
    int32_t identity;  // object transform is the identity
    int32_t kind2;     // media: kind of the boundary shape
    int32_t octree;    // octree index for OCTREE leaves
    int32_t translation;  // the object transform is a pure translation (linear part exactly the identity)
    double p[10];      // shape parameters (media: the boundary's, and p[9] = -1/density)
    Affine to_object;  // world -> object (used when !identity)
};
static_assert(sizeof(DNodeRec) == 256, "DNodeRec must be 256 bytes");

// One self-contained record per object (and per light) for everything that is indexed per lane:
// the leaf test of the per-lane walk, the surface data at a hit, the material response and the
// light sampling. One record means one batch of independent loads instead of the dependent chain
// object -> shape -> (boundary) -> material -> texture.
struct DObjRec {
    int32_t kind;       // shape kind of the object (CONSTANT_MEDIUM for media)
    int32_t gkind;      // kind of the geometry: the boundary's for media, else `kind`
    int32_t identity;   // object transform: 1 the identity, 2 a pure translation (linear part exactly the identity), 0 any other
    int32_t octree;     // octree index for OCTREE shapes
    int32_t mat_kind;   // material kind (phase function for media); unused for OCTREE
    int32_t texture;    // texture index of the material
    int32_t tex_solid;  // 1: the texture is a SolidColor and `color` holds its value
    int32_t pad;
    float color[3];     // SolidColor value, or the Dielectric colour
    int32_t pad2;
    double fuzz, index_refraction;
    double p[10];       // geometry parameters (media: the boundary's, and p[9] = -1/density)
    Affine to_object;   // world_transform
    Affine to_world;    // world_transform^-1
    double normal_m[3][3];
    double pad3;
};
static_assert(sizeof(DObjRec) == 416, "DObjRec layout");

// Host-side flattened scene.
struct FlatScene {
    std::vector<DObject> objects, lights;
    std::vector<DShape> shapes;
    std::vector<DMaterial> materials;
    std::vector<DTexture> textures;
    std::vector<DImage> images;
    std::vector<float> image_pool;
    std::vector<mrt_perlin> perlins;
    std::vector<DOctree> octrees;
    std::vector<uint8_t> octree_has_volume;  // per octree: a Volume voxel makes its hit() stochastic
    std::vector<DOctNode> oct_nodes;
    std::vector<mrt_voxel> oct_voxels;
    DCamera camera;
    mrt_background background;
    mrt_sun sun;

    // BVH in the reference's topology, pre-order (what mrt_scene_bvh exports)
    std::vector<mrt_bvh_node> bvh_full;
    // device BVH: duplicate leaves folded; SoA
    std::vector<double> bmin[3], bmax[3];
    std::vector<int32_t> node_object, node_skip, node_flags;
    std::vector<DNode> nodes;
    std::vector<DLeafRec> lane_leaf;  // per node of `nodes`
    std::vector<DNodeRec> node_recs;
    // The leaves of node_recs in the same (pre-)order, each linked to the next one. A leaf's box lies
    // inside the boxes of its ancestors, so an ancestor's test can only reject rays the leaf's own box
    // rejects too: walking this list gives the results of the tree walk without the inner-node tests,
    // which is cheaper when the rays that share the walk (a wave) have nothing in common.
    // Empty when a leaf box depends on the query (NODE_DYNAMIC_BOX).
    std::vector<DNodeRec> leaf_recs;
    std::vector<DObjRec> obj_recs, light_recs;
    // Candidate search for large trees (see build_candidate_tree): a spatial tree over tight boxes of
    // the leaves of `nodes`. Empty when the scene does not qualify.
    std::vector<DCand4> cand_nodes;
    bool has_medium = false;
};

// Sets the calling thread's mrt_last_error() message (for the library's other translation units).
void set_last_error(const std::string& msg);

// Builds FlatScene from the descriptor. Returns an mrt_status; `err` receives a message.
int flatten_scene(const mrt_scene_desc* desc, FlatScene* out, std::string* err);

}  // namespace mrt
#endif
