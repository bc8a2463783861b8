// mrt_host.cpp -- host side of the scene: validates the descriptor, precomputes
// per-object matrices, builds the BVH with the reference's construction
// (cpu_rt/src/ray_tracer/bvh.rs:206-293) and flattens it into pre-order SoA
// arrays with skip links for stackless traversal on the device.
#include "mrt_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>

namespace mrt {
namespace {

struct Mat4 {
    double m[4][4];  // m[col][row], cgmath layout
};

double det3x3(const double a[3][3]) {  // a[col][row]
    return a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) - a[1][0] * (a[0][1] * a[2][2] - a[2][1] * a[0][2]) +
           a[2][0] * (a[0][1] * a[1][2] - a[1][1] * a[0][2]);
}
double minor(const Mat4& m, int col, int row) {
    double a[3][3];
    int ci = 0;
    for (int c = 0; c < 4; ++c) {
        if (c == col) continue;
        int ri = 0;
        for (int r = 0; r < 4; ++r) {
            if (r == row) continue;
            a[ci][ri++] = m.m[c][r];
        }
        ++ci;
    }
    return det3x3(a);
}
bool invert(const Mat4& m, Mat4* out) {
    double det = 0.0;
    for (int c = 0; c < 4; ++c) det += ((c & 1) ? -1.0 : 1.0) * m.m[c][0] * minor(m, c, 0);
    if (det == 0.0 || !std::isfinite(det)) return false;
    double inv_det = 1.0 / det;
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) {
            double sign = ((col + row) & 1) ? -1.0 : 1.0;
            out->m[col][row] = minor(m, row, col) * sign * inv_det;
        }
    return true;
}
// inverse of the upper-left 3x3 (hittable.rs:206-216): columns cross products / det, transposed
bool invert3(const Mat4& m, double out_rows[3][3]) {
    double c[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[i][j] = m.m[i][j];
    double det = det3x3(c);
    if (det == 0.0 || !std::isfinite(det)) return false;
    auto cross = [](const double* a, const double* b, double* o) {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    };
    double r[3][3];  // r[k] = k-th column of the un-transposed result = k-th ROW of the inverse
    cross(c[1], c[2], r[0]);
    cross(c[2], c[0], r[1]);
    cross(c[0], c[1], r[2]);
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 3; ++j) out_rows[k][j] = r[k][j] / det;
    return true;
}
Affine to_affine(const Mat4& m) {
    Affine a;
    for (int row = 0; row < 3; ++row)
        for (int col = 0; col < 4; ++col) a.r[row][col] = m.m[col][row];
    return a;
}
void apply(const Affine& a, const double p[3], double o[3]) {
    for (int r = 0; r < 3; ++r) o[r] = ((a.r[r][0] * p[0] + a.r[r][1] * p[1]) + a.r[r][2] * p[2]) + a.r[r][3];
}

struct Box {
    double mn[3], mx[3];
};

// Shape bounding boxes (sphere.rs:57-62,175-186; rect.rs:88-93,208-213,333-338; render_box.rs:131-136;
// constant_medium.rs:95-97; fast_oct_tree/hittable.rs:69-79).
bool shape_box(const FlatScene& s, int shape, double t0, double t1, Box* b) {
    const DShape& sh = s.shapes[(size_t)shape];
    const double* p = sh.p;
    switch (sh.kind) {
        case MRT_SHAPE_SPHERE:
            for (int i = 0; i < 3; ++i) b->mn[i] = p[1 + i] - p[0], b->mx[i] = p[1 + i] + p[0];
            return true;
        case MRT_SHAPE_MOVING_SPHERE: {
            double f0 = (t0 - p[7]) / (p[8] - p[7]), f1 = (t1 - p[7]) / (p[8] - p[7]);
            for (int i = 0; i < 3; ++i) {
                double c0 = p[1 + i] + f0 * (p[4 + i] - p[1 + i]);
                double c1 = p[1 + i] + f1 * (p[4 + i] - p[1 + i]);
                double lo0 = c0 - p[0], hi0 = c0 + p[0], lo1 = c1 - p[0], hi1 = c1 + p[0];
                b->mn[i] = lo0 < lo1 ? lo0 : lo1;
                b->mx[i] = hi0 > hi1 ? hi0 : hi1;
            }
            return true;
        }
        case MRT_SHAPE_XY_RECT:
            b->mn[0] = p[0], b->mx[0] = p[1], b->mn[1] = p[2], b->mx[1] = p[3];
            b->mn[2] = p[4] - 0.001, b->mx[2] = p[4] + 0.001;
            return true;
        case MRT_SHAPE_XZ_RECT:
            b->mn[0] = p[0], b->mx[0] = p[1], b->mn[2] = p[2], b->mx[2] = p[3];
            b->mn[1] = p[4] - 0.001, b->mx[1] = p[4] + 0.001;
            return true;
        case MRT_SHAPE_YZ_RECT:
            b->mn[1] = p[0], b->mx[1] = p[1], b->mn[2] = p[2], b->mx[2] = p[3];
            b->mn[0] = p[4] - 0.001, b->mx[0] = p[4] + 0.001;
            return true;
        case MRT_SHAPE_RENDER_BOX:
            for (int i = 0; i < 3; ++i) b->mn[i] = p[i], b->mx[i] = p[3 + i];
            return true;
        case MRT_SHAPE_CONSTANT_MEDIUM: return shape_box(s, sh.boundary, t0, t1, b);
        case MRT_SHAPE_OCTREE: {
            const DOctree& o = s.octrees[(size_t)sh.octree];
            if (o.n_nodes == 0) return false;
            double ws = (double)(1u << o.world_log2);
            for (int i = 0; i < 3; ++i) b->mn[i] = 0.0, b->mx[i] = ws;
            return true;
        }
    }
    return false;
}
// Object::bounding_box (hittable.rs:239-282): only the two corners are transformed.
bool object_box(const FlatScene& s, const DObject& o, double t0, double t1, Box* b) {
    Box sb;
    if (!shape_box(s, o.shape, t0, t1, &sb)) return false;
    double a[3], c[3];
    apply(o.to_world, sb.mn, a);
    apply(o.to_world, sb.mx, c);
    for (int i = 0; i < 3; ++i) {
        b->mn[i] = a[i] < c[i] ? a[i] : c[i];
        b->mx[i] = a[i] > c[i] ? a[i] : c[i];
    }
    return true;
}

uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct TreeNode {
    bool leaf = false;
    size_t object = 0;
    Box box{};
    std::unique_ptr<TreeNode> left, right;
};

struct Builder {
    const FlatScene& s;
    std::vector<Box> box0;  // Object::bounding_box(0, 0) per object, the sort key (bvh.rs:183-196)
    double time0, time1;
    uint64_t rng_state;

    uint32_t next_axis() {  // rand_u32(0, 2), bvh.rs:215
        rng_state += 0x9E3779B97F4A7C15ULL;
        return (uint32_t)(mix64(rng_state) >> 32) % 2u;
    }
    std::unique_ptr<TreeNode> leaf(size_t idx) {
        auto n = std::make_unique<TreeNode>();
        n->leaf = true;
        n->object = idx;
        object_box(s, s.objects[idx], time0, time1, &n->box);
        return n;
    }
    // `order` = the slice of (sorted clones of) objects handed down; leaves take start + offset.
    std::unique_ptr<TreeNode> build(const std::vector<size_t>& order, size_t offset) {
        uint32_t axis = next_axis();
        auto less = [&](size_t a, size_t b) { return box0[a].mn[axis] < box0[b].mn[axis]; };
        size_t span = order.size();
        std::unique_ptr<TreeNode> l, r;
        if (span == 1) {
            l = leaf(offset);
            r = leaf(offset);
        } else if (span == 2) {
            if (less(order[0], order[1])) {
                l = leaf(offset);
                r = leaf(offset + 1);
            } else {
                l = leaf(offset + 1);
                r = leaf(offset);
            }
        } else {
            std::vector<size_t> sorted(order);
            std::stable_sort(sorted.begin(), sorted.end(), less);
            size_t middle = sorted.size() / 2;
            std::vector<size_t> lo(sorted.begin(), sorted.begin() + (long)middle);
            std::vector<size_t> hi(sorted.begin() + (long)middle, sorted.end());
            l = build(lo, offset);
            r = build(hi, offset + middle);
        }
        auto n = std::make_unique<TreeNode>();
        for (int i = 0; i < 3; ++i) {  // surrounding_box, bvh.rs:34-47
            n->box.mn[i] = l->box.mn[i] < r->box.mn[i] ? l->box.mn[i] : r->box.mn[i];
            n->box.mx[i] = l->box.mx[i] > r->box.mx[i] ? l->box.mx[i] : r->box.mx[i];
        }
        n->left = std::move(l);
        n->right = std::move(r);
        return n;
    }
};

// Shapes whose hit() draws random numbers: the two leaves of a span-1 node (bvh.rs:224-232) can then
// answer differently, so the folded node evaluates them twice.
bool shape_is_stochastic(const FlatScene& s, int shape) {
    const DShape& sh = s.shapes[(size_t)shape];
    if (sh.kind == MRT_SHAPE_OCTREE) return s.octree_has_volume[(size_t)sh.octree] != 0;
    return sh.kind == MRT_SHAPE_CONSTANT_MEDIUM;
}
bool shape_is_dynamic(const FlatScene& s, int shape) {
    const DShape& sh = s.shapes[(size_t)shape];
    if (sh.kind == MRT_SHAPE_MOVING_SPHERE) return true;
    if (sh.kind == MRT_SHAPE_CONSTANT_MEDIUM) return shape_is_dynamic(s, sh.boundary);
    return false;
}

void emit_full(const TreeNode* n, std::vector<mrt_bvh_node>* out) {
    size_t me = out->size();
    out->push_back(mrt_bvh_node{});
    mrt_bvh_node rec{};
    for (int i = 0; i < 3; ++i) rec.bmin[i] = n->box.mn[i], rec.bmax[i] = n->box.mx[i];
    rec.object = n->leaf ? (int32_t)n->object : -1;
    if (!n->leaf) {
        emit_full(n->left.get(), out);
        emit_full(n->right.get(), out);
    }
    rec.skip = (int32_t)out->size();
    (*out)[me] = rec;
}

void emit_device(const FlatScene& s, const TreeNode* n, FlatScene* out) {
    size_t me = out->node_object.size();
    auto push = [&](const TreeNode* t, int32_t obj, int32_t flags) {
        for (int i = 0; i < 3; ++i) out->bmin[i].push_back(t->box.mn[i]), out->bmax[i].push_back(t->box.mx[i]);
        out->node_object.push_back(obj);
        out->node_skip.push_back(0);
        out->node_flags.push_back(flags);
    };
    if (n->leaf) {
        int32_t flags = shape_is_dynamic(s, s.objects[n->object].shape) ? NODE_DYNAMIC_BOX : 0;
        push(n, (int32_t)n->object, flags);
    } else if (n->left->leaf && n->right->leaf && n->left->object == n->right->object) {
        // Span-1 node: Child{Leaf k, Leaf k}. The parent's box equals the leaf's box, so the
        // pair collapses into one leaf node. A deterministic shape returns the same record on
        // the second visit (t_max = first t is accepted by the non-strict tests and rejected
        // by the strict octree test, both leaving the first record in place); a stochastic
        // medium draws again, so it is marked to be evaluated twice.
        int shape = s.objects[n->left->object].shape;
        int32_t flags = shape_is_dynamic(s, shape) ? NODE_DYNAMIC_BOX : 0;
        if (flags & NODE_DYNAMIC_BOX) {
            // keep the parent/leaf/leaf structure: a dynamic leaf box differs per visit
            push(n, -1, 0);
            emit_device(s, n->left.get(), out);
            emit_device(s, n->right.get(), out);
        } else {
            if (shape_is_stochastic(s, shape)) flags |= NODE_TWICE;
            push(n->left.get(), (int32_t)n->left->object, flags);
        }
    } else {
        push(n, -1, 0);
        emit_device(s, n->left.get(), out);
        emit_device(s, n->right.get(), out);
    }
    out->node_skip[me] = (int32_t)out->node_object.size();
}

// ---- candidate search for large trees
// The reference's tree splits index ranges of the object list, not space (bvh.rs:254-277 sorts a
// copy, the leaves keep the list positions), so the boxes of a many-object scene overlap almost
// everywhere and a ray visits most of the tree. The walk's result does not need that work: a leaf
// can change the running nearest hit only if the ray meets the leaf's shape. So a lane first
// collects, through a spatial tree over tight world boxes, the leaves whose shape the ray's line
// can meet at all (any order, no nearest-hit bound), marks them and their ancestors, and then
// runs the reference's walk over the marked nodes only: same order, same box tests against the
// same running bound, same leaf code. An unmarked subtree holds no leaf the ray can hit, so
// leaving it out changes nothing. Shapes that draw random numbers (media, volume octrees) number
// their draws by visit, which leaving out visits would change: such scenes do not qualify.
struct CandItem {
    Box a;          // box at ray time 0
    double v[3];    // velocity of the box per unit of ray time
    Box sah;        // box over typical ray times, for the split heuristic only
    int32_t node;
};
// Tight world box of an object as a function of the ray's time tau: a.mn + tau * v .. a.mx + tau * v.
bool tight_box(const FlatScene& s, const DObject& o, CandItem* it) {
    const DShape& sh = s.shapes[(size_t)o.shape];
    Box sb;
    double v[3] = {0.0, 0.0, 0.0};
    if (sh.kind == MRT_SHAPE_MOVING_SPHERE) {
        // centre(tau) = c0 + (tau - t0) / (t1 - t0) * (c1 - c0) = (c0 - t0 * v) + tau * v
        const double* p = sh.p;
        const double dt = p[8] - p[7], r = std::fabs(p[0]);
        for (int i = 0; i < 3; ++i) {
            v[i] = (p[4 + i] - p[1 + i]) / dt;
            double c = p[1 + i] - p[7] * v[i];
            if (!std::isfinite(v[i]) || !std::isfinite(c)) return false;
            sb.mn[i] = c - r, sb.mx[i] = c + r;
        }
    } else {
        if (!shape_box(s, o.shape, 0.0, 0.0, &sb)) return false;
        for (int i = 0; i < 3; ++i)
            if (sb.mn[i] > sb.mx[i]) std::swap(sb.mn[i], sb.mx[i]);  // a negative radius turns the box inside out
    }
    Box& b = it->a;
    for (int i = 0; i < 3; ++i) b.mn[i] = INFINITY, b.mx[i] = -INFINITY;
    for (int k = 0; k < 8; ++k) {
        double c[3] = {(k & 1) ? sb.mx[0] : sb.mn[0], (k & 2) ? sb.mx[1] : sb.mn[1], (k & 4) ? sb.mx[2] : sb.mn[2]}, w[3];
        apply(o.to_world, c, w);
        for (int i = 0; i < 3; ++i) b.mn[i] = std::min(b.mn[i], w[i]), b.mx[i] = std::max(b.mx[i], w[i]);
    }
    for (int i = 0; i < 3; ++i) {  // the linear part of the transform applied to the velocity
        it->v[i] = (o.to_world.r[i][0] * v[0] + o.to_world.r[i][1] * v[1]) + o.to_world.r[i][2] * v[2];
        if (!std::isfinite(b.mn[i]) || !std::isfinite(b.mx[i]) || !std::isfinite(it->v[i])) return false;
        double m = 1e-9 * (1.0 + std::max(std::fabs(b.mn[i]), std::fabs(b.mx[i])));  // margin against rounding
        b.mn[i] -= m, b.mx[i] += m;
        it->sah.mn[i] = b.mn[i], it->sah.mx[i] = b.mx[i];
    }
    return true;
}
// Binary build node (pre-order, skip links); collapsed into four-wide nodes afterwards.
struct Cand2 {
    double lo[3], hi[3], vlo[3], vhi[3];
    int32_t object, skip;
    Box sah;
};
double half_area(const Box& b) {
    double d[3] = {b.mx[0] - b.mn[0], b.mx[1] - b.mn[1], b.mx[2] - b.mn[2]};
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
}
void grow(Box* a, const Box& b) {
    for (int i = 0; i < 3; ++i) a->mn[i] = std::min(a->mn[i], b.mn[i]), a->mx[i] = std::max(a->mx[i], b.mx[i]);
}
const Box kEmptyBox{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
// Surface-area-heuristic split by full sweeps over the three centroid orders; emits pre-order nodes.
void build_candidate_nodes(std::vector<CandItem>& items, size_t lo, size_t hi, std::vector<Cand2>* out) {
    const size_t me = out->size();
    out->push_back(Cand2{});
    Cand2 nd{};
    nd.object = -1;
    nd.sah = kEmptyBox;
    for (size_t i = lo; i < hi; ++i) grow(&nd.sah, items[i].sah);
    for (int a = 0; a < 3; ++a) nd.lo[a] = INFINITY, nd.hi[a] = -INFINITY, nd.vlo[a] = INFINITY, nd.vhi[a] = -INFINITY;
    for (size_t i = lo; i < hi; ++i)
        for (int a = 0; a < 3; ++a) {
            nd.lo[a] = std::min(nd.lo[a], items[i].a.mn[a]), nd.hi[a] = std::max(nd.hi[a], items[i].a.mx[a]);
            nd.vlo[a] = std::min(nd.vlo[a], items[i].v[a]), nd.vhi[a] = std::max(nd.vhi[a], items[i].v[a]);
        }
    for (int a = 0; a < 3; ++a) {
        // widen the velocity interval a little: tau * v is rounded differently from the shape's own centre formula
        nd.vlo[a] -= 1e-9 * std::fabs(nd.vlo[a]), nd.vhi[a] += 1e-9 * std::fabs(nd.vhi[a]);
    }
    if (hi - lo == 1) {
        nd.object = items[lo].node;
    } else {
        const size_t n = hi - lo;
        double best = INFINITY;
        int best_axis = 0;
        size_t best_split = n / 2;
        std::vector<double> right(n);
        auto by_centre = [](int axis) {
            return [axis](const CandItem& x, const CandItem& y) {
                return x.sah.mn[axis] + x.sah.mx[axis] < y.sah.mn[axis] + y.sah.mx[axis];
            };
        };
        for (int axis = 0; axis < 3; ++axis) {
            std::sort(items.begin() + (long)lo, items.begin() + (long)hi, by_centre(axis));
            Box acc = kEmptyBox;
            for (size_t i = n; i-- > 1;) {
                grow(&acc, items[lo + i].sah);
                right[i] = half_area(acc);
            }
            acc = kEmptyBox;
            for (size_t i = 1; i < n; ++i) {
                grow(&acc, items[lo + i - 1].sah);
                double cost = half_area(acc) * (double)i + right[i] * (double)(n - i);
                if (cost < best) best = cost, best_axis = axis, best_split = i;
            }
        }
        std::sort(items.begin() + (long)lo, items.begin() + (long)hi, by_centre(best_axis));
        build_candidate_nodes(items, lo, lo + best_split, out);
        build_candidate_nodes(items, lo + best_split, hi, out);
    }
    nd.skip = (int32_t)out->size();
    (*out)[me] = nd;
}
// Collapses the binary tree below node `i` into a four-wide node; returns its index and the stack
// depth the device traversal needs below it (it descends into the last child that was hit and keeps
// the others on the stack).
int collapse(const std::vector<Cand2>& b2, int i, std::vector<DCand4>* out, int* need) {
    std::vector<int> kids;
    if (b2[(size_t)i].object >= 0) {
        kids.push_back(i);  // a tree of one leaf
    } else {
        kids = {i + 1, b2[(size_t)i + 1].skip};
        while (kids.size() < 4) {
            int pick = -1;
            double area = -1.0;
            for (size_t k = 0; k < kids.size(); ++k)
                if (b2[(size_t)kids[k]].object < 0 && half_area(b2[(size_t)kids[k]].sah) > area)
                    area = half_area(b2[(size_t)kids[k]].sah), pick = (int)k;
            if (pick < 0) break;
            int c = kids[(size_t)pick];
            kids[(size_t)pick] = c + 1;
            kids.push_back(b2[(size_t)c + 1].skip);
        }
    }
    const int me = (int)out->size();
    out->push_back(DCand4{});
    struct Kid {
        int b2, index, need;
    };
    std::vector<Kid> inner, leaves;
    for (int c : kids) {
        if (b2[(size_t)c].object >= 0) {
            leaves.push_back({c, -2 - b2[(size_t)c].object, 0});
        } else {
            int n = 0;
            int idx = collapse(b2, c, out, &n);
            inner.push_back({c, idx, n});
        }
    }
    // the child with the deepest stack demand first: it is entered with the fewest siblings waiting
    std::stable_sort(inner.begin(), inner.end(), [](const Kid& a, const Kid& b) { return a.need > b.need; });
    *need = 0;
    for (size_t j = 0; j < inner.size(); ++j) *need = std::max(*need, (int)j + inner[j].need);
    DCand4 nd{};
    std::vector<Kid> all(inner);
    all.insert(all.end(), leaves.begin(), leaves.end());
    // f32 bounds rounded outward: the stored box contains the exact one
    auto down = [](double x) {
        float f = (float)x;
        return (double)f > x ? std::nextafterf(f, -INFINITY) : f;
    };
    auto up = [](double x) {
        float f = (float)x;
        return (double)f < x ? std::nextafterf(f, INFINITY) : f;
    };
    for (int k = 0; k < 4; ++k) {
        nd.child[k] = -1;
        for (int a = 0; a < 3; ++a) nd.lo[a][k] = INFINITY, nd.hi[a][k] = -INFINITY, nd.vlo[a][k] = 0.f, nd.vhi[a][k] = 0.f;
        if ((size_t)k >= all.size()) continue;
        const Cand2& c = b2[(size_t)all[(size_t)k].b2];
        nd.child[k] = all[(size_t)k].index;
        for (int a = 0; a < 3; ++a) {
            nd.lo[a][k] = down(c.lo[a]), nd.hi[a][k] = up(c.hi[a]);
            nd.vlo[a][k] = down(c.vlo[a]), nd.vhi[a][k] = up(c.vhi[a]);
        }
    }
    (*out)[(size_t)me] = nd;
    return me;
}
void build_candidate_tree(FlatScene* s) {
    if (s->nodes.size() > (size_t)CAND_MAX_NODES) return;
    std::vector<CandItem> items;
    for (size_t i = 0; i < s->nodes.size(); ++i) {
        const DNode& nd = s->nodes[i];
        if (nd.object < 0) continue;
        const DObject& ob = s->objects[(size_t)nd.object];
        if ((nd.flags & NODE_TWICE) || shape_is_stochastic(*s, ob.shape)) return;
        CandItem it;
        if (!tight_box(*s, ob, &it)) return;
        it.node = (int32_t)i;  // index of the leaf in `nodes`
        items.push_back(it);
    }
    if (items.empty()) return;
    // group by where the leaves are for ray times of a few units (bounce distances); the bounds
    // themselves hold for every ray time
    for (CandItem& it : items)
        for (int i = 0; i < 3; ++i) {
            it.sah.mn[i] = it.a.mn[i] + std::min(0.0, 2.0 * it.v[i]);
            it.sah.mx[i] = it.a.mx[i] + std::max(0.0, 2.0 * it.v[i]);
        }
    std::vector<Cand2> b2;
    build_candidate_nodes(items, 0, items.size(), &b2);
    int need = 0;
    collapse(b2, 0, &s->cand_nodes, &need);
    // (the device's queue entries hold the node index in ten bits; `need`, the depth of a lane's own stack, no
    // longer limits anything: the wave's queue overflowing sends its rays to the plain walk)
    if (s->cand_nodes.size() > 1024) {
        s->cand_nodes.clear();
        return;
    }
    bool finite = true;
    for (DCand4& nd : s->cand_nodes) {
        for (int a = 0; a < 3; ++a) nd.bound[a] = CAND_MIN_BOUND;
        nd.vel = 0.f;
        for (int k = 0; k < 4; ++k) {
            if (nd.child[k] == -1) continue;
            for (int a = 0; a < 3; ++a) {
                nd.bound[a] = std::max(nd.bound[a], std::max(std::fabs(nd.lo[a][k]), std::fabs(nd.hi[a][k])));
                nd.vel = std::max(nd.vel, std::max(std::fabs(nd.vlo[a][k]), std::fabs(nd.vhi[a][k])));
                finite = finite && std::isfinite(nd.bound[a]);
            }
        }
        finite = finite && std::isfinite(nd.vel);
    }
    // (bounds too large for the f32 node format end up infinite: no search then)
    if (!finite) s->cand_nodes.clear();
}

int fail(std::string* err, int code, const std::string& msg) {
    if (err) *err = msg;
    return code;
}

}  // namespace

int flatten_scene(const mrt_scene_desc* d, FlatScene* out, std::string* err) {
    if (!d || !out) return fail(err, MRT_ERR_INVALID_ARGUMENT, "null scene descriptor");
    if (d->abi_version != MRT_ABI_VERSION) return fail(err, MRT_ERR_INVALID_ARGUMENT, "ABI version mismatch");
    auto need = [&](uint32_t n, const void* p) { return n == 0 || p != nullptr; };
    if (!need(d->n_textures, d->textures) || !need(d->n_images, d->images) || !need(d->n_materials, d->materials) ||
        !need(d->n_shapes, d->shapes) || !need(d->n_objects, d->objects) || !need(d->n_lights, d->lights) ||
        !need(d->n_octrees, d->octrees) || !need(d->n_perlins, d->perlins))
        return fail(err, MRT_ERR_INVALID_ARGUMENT, "array pointer is NULL with a non-zero count");
    FlatScene& s = *out;
    s = FlatScene{};

    for (uint32_t i = 0; i < d->n_images; ++i) {
        const mrt_image& im = d->images[i];
        if (im.width <= 0 || im.height <= 0 || !im.rgb) return fail(err, MRT_ERR_INVALID_ARGUMENT, "bad image");
        DImage di{im.width, im.height, (uint64_t)s.image_pool.size()};
        s.image_pool.insert(s.image_pool.end(), im.rgb, im.rgb + (size_t)im.width * (size_t)im.height * 3);
        s.images.push_back(di);
    }
    for (uint32_t i = 0; i < d->n_perlins; ++i) {
        const mrt_perlin& pn = d->perlins[i];
        for (int k = 0; k < 256; ++k)
            if (pn.perm_x[k] > 255 || pn.perm_y[k] > 255 || pn.perm_z[k] > 255)
                return fail(err, MRT_ERR_INVALID_ARGUMENT, "perlin: permutation entry out of range");
        s.perlins.push_back(pn);
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const mrt_texture& t = d->textures[i];
        DTexture dt{t.kind, t.a, t.b, t.image, {t.color[0], t.color[1], t.color[2]}, 0};
        switch (t.kind) {
            case MRT_TEX_SOLID:
            case MRT_TEX_DEBUG_V: break;
            case MRT_TEX_IMAGE:
                if (t.image < 0 || (uint32_t)t.image >= d->n_images)
                    return fail(err, MRT_ERR_INVALID_ARGUMENT, "texture: bad image index");
                break;
            case MRT_TEX_MULTIPLY:
            case MRT_TEX_CHECKER:
                if (t.a < 0 || (uint32_t)t.a >= i || t.b < 0 || (uint32_t)t.b >= i)
                    return fail(err, MRT_ERR_INVALID_ARGUMENT, "texture: child index must precede the texture");
                break;
            case MRT_TEX_PERLIN:
                if (t.image < 0 || (uint32_t)t.image >= d->n_perlins)
                    return fail(err, MRT_ERR_INVALID_ARGUMENT, "texture: bad perlin index");
                break;
            default: return fail(err, MRT_ERR_INVALID_ARGUMENT, "unknown texture kind");
        }
        s.textures.push_back(dt);
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const mrt_material& m = d->materials[i];
        if (m.kind < MRT_MAT_LAMBERTIAN || m.kind > MRT_MAT_ISOTROPIC)
            return fail(err, MRT_ERR_INVALID_ARGUMENT, "unknown material kind");
        if (m.kind != MRT_MAT_DIELECTRIC && (m.texture < 0 || (uint32_t)m.texture >= d->n_textures))
            return fail(err, MRT_ERR_INVALID_ARGUMENT, "material: bad texture index");
        s.materials.push_back(DMaterial{m.kind, m.texture, m.fuzz, m.index_refraction,
                                        {m.color[0], m.color[1], m.color[2]}, 0});
    }
    for (uint32_t i = 0; i < d->n_octrees; ++i) {
        const mrt_octree& o = d->octrees[i];
        if (!need(o.n_nodes, o.nodes) || !need(o.n_children, o.children) || !need(o.n_voxels, o.voxels) ||
            (o.n_children % 8) != 0)
            return fail(err, MRT_ERR_INVALID_ARGUMENT, "octree: bad arrays");
        DOctree dt{};
        dt.n_nodes = o.n_nodes;
        const uint32_t voxel_base = (uint32_t)s.oct_voxels.size();
        if (o.n_nodes) {
            if (o.root >= o.n_nodes) return fail(err, MRT_ERR_INVALID_ARGUMENT, "octree: bad root");
            dt.world_log2 = o.nodes[o.root].size;
            if (dt.world_log2 > 20) return fail(err, MRT_ERR_UNSUPPORTED, "octree: world larger than 2^20");
            // breadth-first re-linearisation from the root; validates links on the way
            std::vector<uint32_t> order{o.root};
            std::vector<uint8_t> seen(o.n_nodes, 0);
            seen[o.root] = 1;
            const uint32_t base = (uint32_t)s.oct_nodes.size();
            dt.root = base;
            for (size_t k = 0; k < order.size(); ++k) {
                const mrt_oct_node& n = o.nodes[order[k]];
                DOctNode dn{};
                dn.kind_size = (uint32_t)n.kind | (n.size << 8);
                dn.link = -1;
                if (n.kind == MRT_OCT_PARENT) {
                    if (n.first_child < 0 || (uint32_t)n.first_child + 8 > o.n_children || n.size == 0)
                        return fail(err, MRT_ERR_INVALID_ARGUMENT, "octree: bad parent node");
                    dn.link = (int32_t)(base + order.size());
                    for (int c = 0; c < 8; ++c) {
                        uint32_t ci = o.children[(uint32_t)n.first_child + (uint32_t)c];
                        if (ci >= o.n_nodes || o.nodes[ci].size + 1 != n.size || seen[ci])
                            return fail(err, MRT_ERR_INVALID_ARGUMENT, "octree: bad child link");
                        seen[ci] = 1;
                        order.push_back(ci);
                    }
                } else if (n.kind == MRT_OCT_LEAF) {
                    if (n.voxel < 0 || (uint32_t)n.voxel >= o.n_voxels)
                        return fail(err, MRT_ERR_INVALID_ARGUMENT, "octree: bad voxel index");
                    dn.link = (int32_t)(voxel_base + (uint32_t)n.voxel);
                } else if (n.kind != MRT_OCT_EMPTY) {
                    return fail(err, MRT_ERR_INVALID_ARGUMENT, "octree: unknown node kind");
                }
                s.oct_nodes.push_back(dn);
            }
            // the kinds of a parent's children ride in the parent's record, a leaf's own code in its own
            auto code_of = [&](const DOctNode& c) -> uint32_t {
                const uint32_t k = c.kind_size & 0xFFu;
                if (k != MRT_OCT_LEAF) return k;
                return o.voxels[(uint32_t)c.link - voxel_base].kind == MRT_VOX_VOLUME ? 3u : (uint32_t)MRT_OCT_LEAF;
            };
            for (size_t k = base; k < s.oct_nodes.size(); ++k) {
                DOctNode& dn = s.oct_nodes[k];
                uint32_t kinds = 0;
                if ((dn.kind_size & 0xFFu) == MRT_OCT_PARENT) {
                    for (uint32_t c = 0; c < 8; ++c) kinds |= code_of(s.oct_nodes[(size_t)dn.link + c]) << (2u * c);
                } else {
                    kinds = code_of(dn);
                }
                dn.kind_size = (dn.kind_size & 0xFFFFu) | (kinds << 16);
            }
            dt.cached = (s.oct_nodes[base].kind_size & 0xFFu) == MRT_OCT_PARENT && dt.world_log2 < 16 ? 1u : 0u;
        }
        s.octree_has_volume.push_back(0);
        for (uint32_t v = 0; v < o.n_voxels; ++v) {
            const mrt_voxel& vx = o.voxels[v];
            if (vx.kind == MRT_VOX_VOLUME) s.octree_has_volume.back() = 1;
            if (vx.kind < MRT_VOX_LAMBERTIAN || vx.kind > MRT_VOX_VOLUME)
                return fail(err, MRT_ERR_INVALID_ARGUMENT, "octree: unknown voxel kind");
            if (vx.kind == MRT_VOX_VOLUME) {
                if (vx.edge_kind < MRT_VOX_EDGE_NONE || vx.edge_kind > MRT_VOX_EDGE_REFLECT)
                    return fail(err, MRT_ERR_INVALID_ARGUMENT, "octree: unknown volume edge effect");
                if (!(vx.density > 0.0)) return fail(err, MRT_ERR_INVALID_ARGUMENT, "octree: volume density must be positive");
            }
        }
        s.oct_voxels.insert(s.oct_voxels.end(), o.voxels, o.voxels + o.n_voxels);
        s.octrees.push_back(dt);
    }
    for (uint32_t i = 0; i < d->n_shapes; ++i) {
        const mrt_shape& sh = d->shapes[i];
        DShape ds{sh.kind, sh.material, sh.boundary, sh.octree, {}};
        std::memcpy(ds.p, sh.p, sizeof(ds.p));
        switch (sh.kind) {
            case MRT_SHAPE_SPHERE:
            case MRT_SHAPE_MOVING_SPHERE:
            case MRT_SHAPE_XY_RECT:
            case MRT_SHAPE_XZ_RECT:
            case MRT_SHAPE_YZ_RECT:
            case MRT_SHAPE_RENDER_BOX:
                if (sh.material < 0 || (uint32_t)sh.material >= d->n_materials)
                    return fail(err, MRT_ERR_INVALID_ARGUMENT, "shape: bad material index");
                break;
            case MRT_SHAPE_CONSTANT_MEDIUM: {
                if (sh.material < 0 || (uint32_t)sh.material >= d->n_materials)
                    return fail(err, MRT_ERR_INVALID_ARGUMENT, "medium: bad phase material index");
                if (sh.boundary < 0 || (uint32_t)sh.boundary >= i)
                    return fail(err, MRT_ERR_INVALID_ARGUMENT, "medium: boundary must be an earlier shape");
                int bk = d->shapes[sh.boundary].kind;
                if (bk != MRT_SHAPE_SPHERE && bk != MRT_SHAPE_RENDER_BOX && bk != MRT_SHAPE_XY_RECT &&
                    bk != MRT_SHAPE_XZ_RECT && bk != MRT_SHAPE_YZ_RECT && bk != MRT_SHAPE_MOVING_SPHERE)
                    return fail(err, MRT_ERR_UNSUPPORTED, "medium: unsupported boundary shape");
                if (!(sh.p[0] != 0.0)) return fail(err, MRT_ERR_INVALID_ARGUMENT, "medium: zero density");
                s.has_medium = true;
                ds.p[9] = -1.0 / sh.p[0];  // ConstantMedium::new, constant_medium.rs:27-32
                break;
            }
            case MRT_SHAPE_OCTREE:
                if (sh.octree < 0 || (uint32_t)sh.octree >= d->n_octrees)
                    return fail(err, MRT_ERR_INVALID_ARGUMENT, "shape: bad octree index");
                break;
            default: return fail(err, MRT_ERR_INVALID_ARGUMENT, "unknown shape kind");
        }
        s.shapes.push_back(ds);
    }
    auto mk_object = [&](const mrt_object& o, DObject* out_o, std::string* why) {
        if (o.shape < 0 || (uint32_t)o.shape >= d->n_shapes) {
            *why = "object: bad shape index";
            return MRT_ERR_INVALID_ARGUMENT;
        }
        Mat4 m, inv;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) m.m[c][r] = o.world_transform[c * 4 + r];
        if (!(m.m[0][3] == 0.0 && m.m[1][3] == 0.0 && m.m[2][3] == 0.0 && m.m[3][3] == 1.0)) {
            *why = "object: only affine transforms (last row 0 0 0 1) are supported";
            return MRT_ERR_UNSUPPORTED;
        }
        if (!invert(m, &inv)) {  // hittable.rs:69-76 expect("transform is not invertible")
            *why = "object: transform is not invertible";
            return MRT_ERR_INVALID_ARGUMENT;
        }
        out_o->shape = o.shape;
        bool ident = true;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) ident = ident && (m.m[c][r] == (c == r ? 1.0 : 0.0));
        out_o->identity = ident ? 1 : 0;
        out_o->to_object = to_affine(m);
        out_o->to_world = to_affine(inv);
        if (!invert3(m, out_o->normal_m)) {
            *why = "object: transform is not invertible";
            return MRT_ERR_INVALID_ARGUMENT;
        }
        return MRT_OK;
    };
    s.objects.resize(d->n_objects);
    for (uint32_t i = 0; i < d->n_objects; ++i) {
        std::string why;
        int rc = mk_object(d->objects[i], &s.objects[i], &why);
        if (rc != MRT_OK) return fail(err, rc, why);
    }
    s.lights.resize(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        std::string why;
        int rc = mk_object(d->lights[i], &s.lights[i], &why);
        if (rc != MRT_OK) return fail(err, rc, why);
        // sphere.rs:187-192 and fast_oct_tree/hittable.rs:80-85 are todo!() in the reference
        int k = s.shapes[(size_t)s.lights[i].shape].kind;
        if (k == MRT_SHAPE_CONSTANT_MEDIUM) k = s.shapes[(size_t)s.shapes[(size_t)s.lights[i].shape].boundary].kind;
        if (k == MRT_SHAPE_MOVING_SPHERE || k == MRT_SHAPE_OCTREE)
            return fail(err, MRT_ERR_UNSUPPORTED, "light: MovingSphere / octree lights panic in the reference");
    }

    // per-object / per-light records for the device
    auto mk_rec = [&](const DObject& ob) {
        DObjRec r;
        std::memset(&r, 0, sizeof(r));
        const DShape& sh = s.shapes[(size_t)ob.shape];
        const DShape& geo = sh.kind == MRT_SHAPE_CONSTANT_MEDIUM ? s.shapes[(size_t)sh.boundary] : sh;
        r.kind = sh.kind, r.gkind = geo.kind, r.identity = ob.identity, r.octree = sh.octree;
        std::memcpy(r.p, geo.p, sizeof(r.p));
        if (sh.kind == MRT_SHAPE_CONSTANT_MEDIUM) r.p[9] = sh.p[9];
        r.texture = -1;
        if (sh.kind != MRT_SHAPE_OCTREE) {
            const DMaterial& m = s.materials[(size_t)sh.material];
            r.mat_kind = m.kind, r.texture = m.texture, r.fuzz = m.fuzz, r.index_refraction = m.index_refraction;
            std::memcpy(r.color, m.color, sizeof(r.color));
            if (m.kind != MRT_MAT_DIELECTRIC && m.texture >= 0 && s.textures[(size_t)m.texture].kind == MRT_TEX_SOLID) {
                r.tex_solid = 1;
                std::memcpy(r.color, s.textures[(size_t)m.texture].color, sizeof(r.color));
            }
        }
        r.to_object = ob.to_object, r.to_world = ob.to_world;
        std::memcpy(r.normal_m, ob.normal_m, sizeof(r.normal_m));
        if (!ob.identity) {
            r.identity = 2;  // a pure translation, unless the linear part says otherwise
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b)
                    if (ob.to_object.r[a][b] != (a == b ? 1.0 : 0.0)) r.identity = 0;
        }
        return r;
    };
    for (const DObject& ob : s.objects) s.obj_recs.push_back(mk_rec(ob));
    for (const DObject& ob : s.lights) s.light_recs.push_back(mk_rec(ob));

    // Camera::new (camera.rs:44-76)
    {
        const mrt_camera_info& ci = d->camera;
        double theta = ci.fov * M_PI / 180.0;
        double h = std::tan(theta / 2.0);
        double world_height = 2.0 * h;
        double world_width = ci.aspect_ratio * world_height;
        auto norm = [](double* v) {
            double s = 1.0 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            v[0] *= s, v[1] *= s, v[2] *= s;
        };
        auto cross = [](const double* a, const double* b, double* o) {
            o[0] = a[1] * b[2] - a[2] * b[1];
            o[1] = a[2] * b[0] - a[0] * b[2];
            o[2] = a[0] * b[1] - a[1] * b[0];
        };
        double w[3], u[3], v[3];
        for (int i = 0; i < 3; ++i) w[i] = ci.origin[i] - ci.look_at[i];
        norm(w);
        cross(ci.up_vector, w, u);
        norm(u);
        cross(w, u, v);
        DCamera& c = s.camera;
        for (int i = 0; i < 3; ++i) {
            c.origin[i] = ci.origin[i];
            c.u[i] = u[i];
            c.v[i] = v[i];
            c.horizontal[i] = ci.focus_distance * world_width * u[i];
            c.vertical[i] = ci.focus_distance * world_height * v[i];
            c.lower_left_corner[i] = ci.origin[i] - c.horizontal[i] / 2.0 - c.vertical[i] / 2.0 - ci.focus_distance * w[i];
        }
        c.lens_radius = ci.aperture / 2.0;
        c.start_time = ci.start_time;
        c.end_time = ci.end_time;
        for (int i = 0; i < 3; ++i)
            if (!std::isfinite(c.lower_left_corner[i]) || !std::isfinite(c.horizontal[i]) || !std::isfinite(c.vertical[i]))
                return fail(err, MRT_ERR_INVALID_ARGUMENT, "camera: degenerate frame");
    }
    if (d->background.kind < MRT_BG_SKY || d->background.kind > MRT_BG_CONSTANT)
        return fail(err, MRT_ERR_INVALID_ARGUMENT, "unknown background kind");
    s.background = d->background;
    s.sun = d->sun;
    if (s.sun.has_sun) {
        // pdf.rs:176-179: panics when the sun direction is (anti)parallel to +z
        double r = std::cos(s.sun.phi);
        double sx = r * std::cos(s.sun.theta), sy = std::sin(s.sun.phi);
        if (std::sqrt(sx * sx + sy * sy) < 0.01)
            return fail(err, MRT_ERR_UNSUPPORTED, "sun direction parallel to +z panics in the reference");
    }

    // BvhTree::new (bvh.rs:78-96)
    if (!s.objects.empty()) {
        Builder b{s, {}, d->camera.start_time, d->camera.end_time, d->bvh_seed};
        b.box0.resize(s.objects.size());
        for (size_t i = 0; i < s.objects.size(); ++i)
            if (!object_box(s, s.objects[i], 0.0, 0.0, &b.box0[i]))
                return fail(err, MRT_ERR_INVALID_ARGUMENT, "object has no bounding box (empty octree)");
        std::vector<size_t> order(s.objects.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        auto root = b.build(order, 0);
        emit_full(root.get(), &s.bvh_full);
        emit_device(s, root.get(), &s);
        s.nodes.resize(s.node_object.size());
        for (size_t i = 0; i < s.nodes.size(); ++i) {
            DNode& nd = s.nodes[i];
            for (int a = 0; a < 3; ++a) nd.bmin[a] = s.bmin[a][i], nd.bmax[a] = s.bmax[a][i];
            nd.object = s.node_object[i], nd.skip = s.node_skip[i], nd.flags = s.node_flags[i], nd.parent = -1;
        }
        {
            // parents in the pre-order array (a node's subtree ends at its skip link)
            std::vector<int32_t> open;
            for (size_t i = 0; i < s.nodes.size(); ++i) {
                while (!open.empty() && s.nodes[(size_t)open.back()].skip <= (int32_t)i) open.pop_back();
                s.nodes[i].parent = open.empty() ? -1 : open.back();
                if (s.nodes[i].object < 0) open.push_back((int32_t)i);
            }
        }
        // per-node leaf records of the per-lane walk (same index as `nodes`)
        s.lane_leaf.assign(s.nodes.size(), DLeafRec{});
        for (size_t i = 0; i < s.nodes.size(); ++i) {
            if (s.nodes[i].object < 0) continue;
            const DObjRec& r = s.obj_recs[(size_t)s.nodes[i].object];
            DLeafRec& l = s.lane_leaf[i];
            bool translation = true;  // linear part exactly the identity
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) translation = translation && r.to_object.r[a][b] == (a == b ? 1.0 : 0.0);
            l.kind = r.kind, l.gkind = r.gkind, l.octree = r.octree;
            l.identity = r.identity == 1 ? 1 : (translation ? 2 : 0);
            std::memcpy(l.p, r.p, sizeof(l.p));
        }
        // self-contained node records for the wave-synchronous walk
        s.node_recs.resize(s.node_object.size());
        for (size_t i = 0; i < s.node_recs.size(); ++i) {
            DNodeRec& r = s.node_recs[i];
            std::memset(&r, 0, sizeof(r));
            for (int a = 0; a < 3; ++a) r.bmin[a] = s.bmin[a][i], r.bmax[a] = s.bmax[a][i];
            r.object = s.node_object[i];
            r.skip = s.node_skip[i];
            r.flags = s.node_flags[i];
            r.kind = -1;
            if (r.object >= 0) {
                const DObject& ob = s.objects[(size_t)r.object];
                const DShape* sh = &s.shapes[(size_t)ob.shape];
                r.kind = sh->kind;
                r.identity = ob.identity;
                r.translation = 1;
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b)
                        if (ob.to_object.r[a][b] != (a == b ? 1.0 : 0.0)) r.translation = 0;
                r.octree = sh->octree;
                r.to_object = ob.to_object;
                if (sh->kind == MRT_SHAPE_CONSTANT_MEDIUM) {
                    const DShape* b = &s.shapes[(size_t)sh->boundary];
                    r.kind2 = b->kind;
                    std::memcpy(r.p, b->p, sizeof(r.p));
                    r.p[9] = sh->p[9];  // -1/density (a boundary uses at most slots 0..8)
                } else {
                    std::memcpy(r.p, sh->p, sizeof(r.p));
                }
                if (r.identity) r.flags |= NODE_IDENTITY;
                if (r.translation) r.flags |= NODE_TRANSLATION;
                r.flags |= (r.kind2 & 0xFF) << NODE_KIND2_SHIFT;
                if (ob.identity && (sh->kind == MRT_SHAPE_XY_RECT || sh->kind == MRT_SHAPE_XZ_RECT || sh->kind == MRT_SHAPE_YZ_RECT ||
                                    sh->kind == MRT_SHAPE_SPHERE))
                    r.flags |= NODE_BOX_IMPLIED;
            }
        }
        build_candidate_tree(&s);
        bool dynamic = false;
        for (const DNodeRec& r : s.node_recs) dynamic = dynamic || (r.flags & NODE_DYNAMIC_BOX);
        if (!dynamic) {
            for (const DNodeRec& r : s.node_recs)
                if (r.object >= 0) {
                    s.leaf_recs.push_back(r);
                    s.leaf_recs.back().skip = (int32_t)s.leaf_recs.size();
                }
        }
    }
    return MRT_OK;
}

}  // namespace mrt
