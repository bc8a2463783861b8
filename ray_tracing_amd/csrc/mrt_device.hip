// mrt_device.hip -- gfx950 wavefront path tracer behind the C ABI of include/mrt.h.
//
// Execution model: a pool of path slots lives in HBM as structure-of-arrays
// (ray origin/direction/time as seven f64 streams, throughput as three f32
// streams, pixel, RNG key, segment counter). One "segment kernel" launch
// advances every live slot by up to `segments_per_launch` path segments
// (nearest-hit query + material response). A slot whose path ended claims the
// next pixel-sample from a global work counter (wave-aggregated: ballot +
// popcount + one atomic per wave) and regenerates a camera ray in place, so
// waves stay full until the work runs out; after that the survivors are packed
// to the front of the pool by a ballot/prefix-sum compaction kernel.
//
// Every device function cites the reference lines it implements
// (paths relative to cpu_rt/src/).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is bound with dlopen at first use

#include <dlfcn.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "mrt_internal.h"

namespace mrt {

// ============================================================== device-side views
struct SceneView {
    const DNode* nodes;  // the tree for the per-lane walk
    const DLeafRec* lane_leaf;  // per node: the leaf's shape record (per-lane walk)
    const DNodeRec* node_recs;  // what the wave-synchronous walk visits: the tree, or its leaf list
    int32_t n_wave_nodes;
    int32_t n_nodes;
    int32_t n_objects;
    int32_t n_lights;
    int32_t trace_zero_throughput;
    int32_t n_cand_nodes;  // > 0: the per-lane query runs the candidate search instead of the walk
    int32_t n_cand_words;  // 32-node words of the candidate bit array in use
    int32_t n_perlins;
    int32_t cand_queue;    // entries of a wave's search queue in use (CAND_QUEUE; smaller only to test the overflow path)
    int32_t oct_first;     // 1: the first node the wave-synchronous walk visits is a leaf holding a solid-voxel octree whose walk can be suspended (octree_first_leaf)
    int32_t pad_view;
    const DCand4* cand_nodes;
    const DObjRec* objects;  // one record per object (shape, transform, material)
    const DObjRec* lights;   // the light list, same record type
    const DTexture* textures;
    const DImage* images;
    const float* image_pool;
    const mrt_perlin* perlins;
    const DOctree* octrees;
    const DOctNode* oct_nodes;
    const mrt_voxel* oct_voxels;
    DCamera cam;
    mrt_background bg;
    mrt_sun sun;
    // constants of the scene that the reference works out per call with sin / cos (evaluated once, on the host):
    double bg_sun_ray[3], bg_cos_sun_radius;  // SunSky::color, sun_sky.rs:32-60
    double sun_dir[3], sun_cos_radius;        // Sun::make_direction_vector, sun.rs:17-20; SkyPdf::value, pdf.rs:135-148
};
#define CONST_AS __attribute__((address_space(4)))
#define LDS_AS __attribute__((address_space(3)))
template <class T>
__device__ __forceinline__ const CONST_AS T* as_const(const T* p) {
    return (const CONST_AS T*)p;
}
// The kernels get the scene view as a pointer to a copy in device memory and read it through the constant
// address space: a field is a scalar load where it is used. Passed by value, its 110 dwords were loaded into
// SGPRs at kernel entry, stayed live across the whole segment loop and were spilled into VGPR lanes, so that
// every use cost vector instructions (v_readlane) to get the value back.
using SceneRef = const CONST_AS SceneView&;

// The path pool: ONE block of memory holding, for `cap` slots, seven f64 arrays (ray origin, direction,
// time), the RNG keys, three f32 arrays (throughput), the pixel indices and the segment counters. A kernel
// gets the block's address and `cap` (three SGPRs instead of thirteen array pointers that would stay live
// across the whole segment loop) and forms an array's address where it is used.
struct Pool {
    unsigned char* base;
    unsigned long long cap;  // slots per array, a multiple of 64
    __host__ __device__ double* f64(int k) const { return reinterpret_cast<double*>(base) + (size_t)k * cap; }
    __host__ __device__ double* ox() const { return f64(0); }
    __host__ __device__ double* oy() const { return f64(1); }
    __host__ __device__ double* oz() const { return f64(2); }
    __host__ __device__ double* dx() const { return f64(3); }
    __host__ __device__ double* dy() const { return f64(4); }
    __host__ __device__ double* dz() const { return f64(5); }
    __host__ __device__ double* tm() const { return f64(6); }
    __host__ __device__ uint64_t* key() const { return reinterpret_cast<uint64_t*>(base) + (size_t)7 * cap; }
    __host__ __device__ float* f32(int k) const { return reinterpret_cast<float*>(base + (size_t)64 * cap) + (size_t)k * cap; }
    __host__ __device__ float* tr() const { return f32(0); }
    __host__ __device__ float* tg() const { return f32(1); }
    __host__ __device__ float* tb() const { return f32(2); }
    __host__ __device__ uint32_t* pixel() const { return reinterpret_cast<uint32_t*>(base + (size_t)76 * cap); }
    // >= 0: index of the next segment of a live path; < 0: free slot
    __host__ __device__ int32_t* seg() const { return reinterpret_cast<int32_t*>(base + (size_t)80 * cap); }
    static constexpr size_t bytes(size_t cap) { return (size_t)84 * cap; }
};
// Work items (pixel-samples) are handed out in chunks: a wave takes WORK_CHUNK consecutive items
// from the global counter with one atomic and serves its lanes from that range; what is left of
// the range when a launch ends is kept per wave ([next, end) pairs) for the next launch.
constexpr uint32_t WORK_CHUNK = 1024;
constexpr int BLOCK = 256;  // threads per workgroup of every kernel here

enum : int { ST_REGEN = 0, ST_TRAVERSE, ST_LEAF, ST_SURFACE, ST_LAMBERT_GEN, ST_LIGHT_VALUE, ST_SHADE, ST_ACCUM, ST_TOTAL, ST_MEDIUM_TAIL, ST_CLAIM, ST_GEN_COS, ST_GEN_LIGHT, ST_GEN_SKY, ST_UNIT_VEC, ST_CAND_SEARCH, ST_COUNT };
struct Counters {
    unsigned long long work_next;
    unsigned long long rays;
    unsigned long long light_queries;
    unsigned int live;
    unsigned int compacted;
    unsigned long long held;  // work items claimed by waves but not started when the launch ended
    unsigned long long paths;  // paths that ended (their radiance added to the image), counted where they end
    unsigned long long stamps[ST_COUNT];  // diagnostic build only
    unsigned long long stamp_lanes[ST_COUNT];
    unsigned long long stamp_counts[12];
};

// Diagnostic build (-DMRT_STAMPS): wave-level cycle shares of the segment kernel's regions.
// A region's cycles are added by the first active lane only, so the sum over lanes and waves is
// the total wave time spent in the region. No stamp executes in the regular build.
#ifdef MRT_STAMPS
struct Stamps {
    unsigned long long acc[ST_COUNT];
    unsigned long long lanes[ST_COUNT];  // cycles x active lanes (lane utilisation of the region)
    unsigned int count[12];              // event counts: candidate search 0-3 per lane, 4-6 per wave (rounds, queries); octree walk 7 steps per lane, 8 step rounds per wave, 9 walks per wave, 10 walks per lane
};
#define STAMP_COUNT(st, i) ((st).count[i]++)
#define STAMP_COUNT_WAVE(st, i) ((st).count[i] += (__lane_id() == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) ? 1u : 0u)
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ __forceinline__ void stamp_add(Stamps& st, int i, unsigned long long t0) {
    unsigned long long d = stamp_now() - t0;
    unsigned long long m = __ballot(1);
    if ((int)__lane_id() == __ffsll((long long)m) - 1) st.acc[i] += d, st.lanes[i] += d * (unsigned long long)__popcll(m);
}
#define REGION_BEGIN(name) unsigned long long name##_t0 = stamp_now()
#define REGION_END(st, i, name) stamp_add(st, i, name##_t0)
#else
struct Stamps {};
#define STAMP_COUNT(st, i) ((void)0)
#define STAMP_COUNT_WAVE(st, i) ((void)0)
#define REGION_BEGIN(name) ((void)0)
#define REGION_END(st, i, name) ((void)0)
#endif

struct RenderParams {
    float* accum;
    uint32_t width, height, x0, strip_w;
    unsigned long long n_work;      // strip_w * height * spp
    unsigned long long npix_strip;  // strip_w * height
    uint64_t seed, first_sample;
    uint32_t n_slots;
    uint32_t steps;
    uint32_t chunk;  // work items a wave claims at a time
    uint32_t all_claimed;  // the global counter had run past the end of the work before this launch
    double inv_npix_strip, inv_strip_w;  // reciprocals for the work-index decomposition
    double inv_w1, inv_h1;               // 1 / (width - 1), 1 / (height - 1) for camera_ray; NaN: divide
};

// w -> (w / d, w % d) for w < 2^53, d < 2^32 and a quotient < 2^32, without an integer division:
// the f64 estimate is off by at most one, which the remainder check repairs.
__device__ __forceinline__ void divmod_u32(unsigned long long w, uint32_t d, double inv_d, uint32_t& q, uint32_t& r) {
    uint32_t qe = (uint32_t)((double)w * inv_d);
    long long rem = (long long)(w - (unsigned long long)qe * d);
    if (rem < 0) qe--, rem += d;
    else if (rem >= (long long)d) qe++, rem -= d;
    q = qe, r = (uint32_t)rem;
}

// Compile-time scene features. The segment kernel is instantiated for a handful of feature sets
// and a scene runs on the smallest instantiation that covers what it contains, so a scene pays
// (registers, instruction cache) only for the shapes, materials and backgrounds it uses.
enum : int {
    FEAT_OCTREE = 1,    // OctTree shapes (voxel walk)
    FEAT_SPHERE = 2,    // Sphere / MovingSphere shapes (objects, medium boundaries, lights)
    FEAT_MEDIUM = 4,    // ConstantMedium shapes
    FEAT_SPECULAR = 8,  // Metal / Dielectric materials
    FEAT_TEXTURE = 16,  // textures other than SolidColor
    FEAT_SKY = 32,      // Sky / SunSky background, or a sun in the sky pdf
    FEAT_PERLIN = 64,   // Perlin textures (implies FEAT_TEXTURE)
    FEAT_VOLUME = 128,  // volume / reflecting voxels in an octree (implies FEAT_OCTREE)
    FEAT_MOVING = 256,  // MovingSphere shapes (implies FEAT_SPHERE): centre by ray time, leaf boxes that depend on the query
    FEAT_SPHERE_LEAF = 512,  // an untransformed Sphere as an object of its own (implies FEAT_SPHERE): direct leaf test in the wave walk
    FEAT_ALL = 1023
};

// ============================================================== small vector math
struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ D3 operator/(D3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double magnitude(D3 a) { return sqrt(dot(a, a)); }
// 1/x from v_rcp_f64 plus two Newton steps (about 1 ulp; an IEEE divide costs three times the
// instructions). Zero, infinity and NaN keep the hardware result (inf, 0, NaN), which the slab
// and plane tests rely on.
__device__ __forceinline__ double rcp_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    double r1 = fma(r, e, r);
    e = fma(-x, r1, 1.0);
    r1 = fma(r1, e, r1);
    // (whenever the hardware result is infinite, zero or NaN the refinement is NaN: 0 * inf in the residual,
    // inf - inf in the update)
    return r1 == r1 ? r1 : r;
}
// a / d, correctly rounded, from inv = 1 / d rounded to nearest (Markstein's division): the product is within
// two units of the last place of a / d, one correction by the exact residual makes it faithful, and a second one
// on a faithful quotient gives the correctly rounded one -- seven instructions for the thirty of a division.
// Holds for d a normal number of moderate size whose significand is not all ones, and a zero or far from the
// ends of the exponent range; NaN stays NaN.
__device__ __forceinline__ double div_by_reciprocal(double a, double d, double inv) {
#pragma clang fp contract(off)
    const double q0 = a * inv;
    const double q1 = fma(fma(-d, q0, a), inv, q0);
    return fma(fma(-d, q1, a), inv, q1);
}
// a / d for any finite a from inv = rcp_fast(d) (within an ulp of 1 / d): the product and two corrections by the exact
// residual. The first leaves a faithful quotient, the second the correctly rounded one unless a / d lies within
// 2^-52 ulp of a rounding boundary, which a quotient of two doubles comes to with a probability of that order. A zero
// divisor (inv infinite) keeps the product, the IEEE quotient there (+-inf, or NaN for 0 / 0), of which the corrections
// would make NaN. Eight instructions for the thirty of the division.
__device__ __forceinline__ double div_by_rcp(double a, double d, double inv) {
#pragma clang fp contract(off)
    const double q0 = a * inv;
    const double q1 = fma(fma(-d, q0, a), inv, q0);
    const double q2 = fma(fma(-d, q1, a), inv, q1);
    return q2 == q2 ? q2 : q0;
}
// 1/sqrt(x) from v_rsq_f64 plus two Newton steps, for x > 0 finite.
__device__ __forceinline__ double rsqrt_fast(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}
// cgmath normalize: self * (1 / magnitude)
__device__ __forceinline__ D3 normalize(D3 a) {
    double d2 = dot(a, a);
    double inv = rsqrt_fast(d2);
    if (__builtin_expect(!(d2 > 1e-300 && d2 < 1e300), 0)) inv = 1.0 / sqrt(d2);  // zero, huge, NaN: IEEE path
    return a * inv;
}
// Natural logarithm of a uniform draw u = k * 2^-53, 0 <= u < 1 (the free-path draw of a medium,
// constant_medium.rs:74-75), by the classic argument reduction u = m * 2^e with m in [sqrt(1/2), sqrt(2)),
// s = (m - 1) / (m + 1), log m = 2 atanh s as a degree-7 polynomial in s^2 (the fdlibm kernel and coefficients;
// error below 1 ulp with an exact quotient, about 1.5 ulp with the refined reciprocal used here). 37 instructions
// for the library routine's 101, which works in double-double to stay below 1 ulp for every argument; a medium
// visit draws once and a Cornell smoke segment holds three visits. log(0) = -inf.
__device__ __forceinline__ double log_unit(double u) {
    double m = __builtin_amdgcn_frexp_mant(u);  // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(u);
    const bool low = m < 0.70710678118654752;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double d = 2.0 + f;  // in [1.7, 2.42): no special cases
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    const double sq = f * r;
    const double dk = (double)e;
    const double z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double big_r = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double res = dk * 6.93147180369123816490e-01 - ((hfsq - fma(sq, hfsq + big_r, dk * 1.90821492927058770002e-10)) - f);
    return u == 0.0 ? -INFINITY : res;
}
constexpr double INV_PI = 0.31830988618379067154;
__device__ __forceinline__ double comp(D3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
template <class P>
__device__ __forceinline__ D3 ld3(P p) { return {p[0], p[1], p[2]}; }

struct RayD {
    D3 o, d;
    double time;
};
__device__ __forceinline__ D3 ray_at(const RayD& r, double t) { return r.o + t * r.d; }  // prelude.rs:161-165

struct F3 {
    float r, g, b;
};
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }

// fmin / fmax as single instructions. v_min_f64 / v_max_f64 return the other operand when one is NaN and quiet
// a signalling NaN by themselves, which is fmin / fmax; the compiler's own lowering first passes every operand it
// cannot prove quiet through another v_max_f64 (x, x) -- seven of the forty instructions of a box test.
__device__ __forceinline__ double vmin_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ============================================================== counter-based RNG (contract: include/mrt.h)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t path_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
    return mix64(mix64(seed + 0x9E3779B97F4A7C15ULL * (pixel + 1)) ^ (0xD1B54A32D192ED03ULL * (sample + 1)));
}
// u64(key, seg, dim) = mix64(key + G * (((seg << 16) | dim) + 1)) with dim < 2^16. The part that
// depends on the segment, key + G * (seg << 16), is formed once per segment; a draw then adds the
// constant G * (dim + 1) (a compile-time constant for nearly every dimension) and mixes.
struct Rng {
    uint64_t base;
    __device__ __forceinline__ Rng(uint64_t key, uint32_t seg) : base(key + 0x9E3779B97F4A7C15ULL * ((uint64_t)seg << 16)) {}
};
__device__ __forceinline__ uint64_t rng_u64(const Rng& g, uint32_t dim) {
    return mix64(g.base + 0x9E3779B97F4A7C15ULL * (uint64_t)(dim + 1));
}
// G * (dim + 1), the per-dimension addend of rng_u64, as a value (lets the caller select between
// compile-time constants per lane instead of multiplying at run time)
constexpr uint64_t rng_dim_addend(uint32_t dim) { return 0x9E3779B97F4A7C15ULL * (uint64_t)(dim + 1); }
__device__ __forceinline__ double rng_f64_addend(const Rng& g, uint64_t addend) {
    return (double)(mix64(g.base + addend) >> 11) * 0x1.0p-53;
}
__device__ __forceinline__ double rng_f64(const Rng& g, uint32_t dim) {
    return (double)(rng_u64(g, dim) >> 11) * 0x1.0p-53;
}
// prelude.rs:82-84
__device__ __forceinline__ double rand_scalar(const Rng& g, double mn, double mx, uint32_t dim) {
    return rng_f64(g, dim) * (mx - mn) + mn;
}
// prelude.rs:85-87
__device__ __forceinline__ uint32_t rand_u32(const Rng& g, uint32_t mn, uint32_t mx, uint32_t dim) {
    return ((uint32_t)(rng_u64(g, dim) >> 32) % (mx - mn)) + mn;
}
// ray_tracer.rs:48-63: rejection sampling inside the unit ball, not normalised
// (u0, u1): the first two uniforms of the first candidate when the caller has drawn them already
// (dimensions base and base + 1), else negative.
__device__ D3 rand_unit_vec(const Rng& g, uint32_t base, double u0 = -1.0, double u1 = -1.0) {
#pragma nounroll
    for (uint32_t k = 0; k < MRT_MAX_REJECT; ++k) {
        const bool pre = k == 0 && u0 >= 0.0;
        D3 rv{pre ? u0 : rng_f64(g, base + 3 * k), pre ? u1 : rng_f64(g, base + 3 * k + 1), rng_f64(g, base + 3 * k + 2)};
        D3 v = 2.0 * (rv - D3{0.5, 0.5, 0.5});
        if (dot(v, v) < 1.0) return v;
    }
    return {0.0, 0.0, 0.0};
}
// rand_unit_vec for the lanes of a wave that ask for one (`want`, from the Rng base and the dimension base
// they pass), worked out by the whole wave. Alone, a lane makes 1.9 attempts on average and the wave waits
// for its unluckiest lane (about five rounds of three draws with a dozen lanes asking) while the other
// lanes idle. Here the 64 lanes split into groups of 1, 2, 4 or 8 per request; lane a of a group tries
// attempt first + a of that request (the draws are addressed by dimension, so any lane can make them), the
// request takes the first attempt of its group that lands in the ball, and requests whose group failed go
// round again with the next attempts. Same vector as rand_unit_vec, bit for bit. Every lane of the wave
// must call this together.
// DIMS = 3: the unit ball (rand_unit_vec). DIMS = 2: the unit disk of the camera's lens (random_in_unit_disk,
// camera.rs:99-106: rand_scalar(-1, 1) twice per attempt, z = 0), attempt k at dimensions dim_base + 2k, + 1.
template <int DIMS>
__device__ D3 wave_rand_reject(bool want, uint64_t rng_base, uint32_t dim_base) {
    D3 out{0.0, 0.0, 0.0};
    const uint32_t lane = __lane_id();
    uint32_t first = 0;  // attempts of this lane's request that have failed
    unsigned long long mask = __ballot(want);
    if (__popcll(mask) > 32) {
        // more than half of the lanes ask: a round of the scheme below would give every request one attempt, made by
        // some other lane; each asking lane makes its own first attempt instead, with nothing to carry between lanes
        const uint64_t at = rng_base + 0x9E3779B97F4A7C15ULL * (uint64_t)(dim_base + 1u);
        D3 v;
        if constexpr (DIMS == 3) {
            D3 rv{(double)(mix64(at) >> 11) * 0x1.0p-53, (double)(mix64(at + 0x9E3779B97F4A7C15ULL) >> 11) * 0x1.0p-53,
                  (double)(mix64(at + 2u * 0x9E3779B97F4A7C15ULL) >> 11) * 0x1.0p-53};
            v = 2.0 * (rv - D3{0.5, 0.5, 0.5});
        } else {
            v = D3{fma((double)(mix64(at) >> 11) * 0x1.0p-53, 2.0, -1.0), fma((double)(mix64(at + 0x9E3779B97F4A7C15ULL) >> 11) * 0x1.0p-53, 2.0, -1.0), 0.0};
        }
        if (want) {
            if (dot(v, v) < 1.0) out = v, want = false;
            else first = 1;
        }
        mask = __ballot(want);
    }
    while (mask != 0) {
        const uint32_t n = (uint32_t)__popcll(mask);
        const uint32_t log_a = n > 32 ? 0u : (n > 16 ? 1u : (n > 8 ? 2u : 3u));  // attempts per request this round
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        // lane q < n learns which lane holds the request of rank q (a permutation of the lanes: the
        // lanes without a request fill the places from n on)
        const uint32_t place = want ? rank : n + lane - rank;
        const int holder = __builtin_amdgcn_ds_permute((int)(place << 2), (int)lane);
        const uint32_t q = lane >> log_a, a = lane & ((1u << log_a) - 1u);
        const int from = __builtin_amdgcn_ds_bpermute((int)(q << 2), holder) << 2;  // (some lane without a request for q >= n)
        const uint32_t b_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)rng_base);
        const uint32_t b_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)(rng_base >> 32));
        const uint32_t r_dim = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)dim_base);
        const uint32_t r_first = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)first);
        const uint32_t attempt = r_first + a;
        const uint64_t at = (((uint64_t)b_hi << 32) | b_lo) + 0x9E3779B97F4A7C15ULL * (uint64_t)(r_dim + (uint32_t)DIMS * attempt + 1u);
        D3 v;
        bool inside;
        if constexpr (DIMS == 3) {
            D3 rv{(double)(mix64(at) >> 11) * 0x1.0p-53, (double)(mix64(at + 0x9E3779B97F4A7C15ULL) >> 11) * 0x1.0p-53,
                  (double)(mix64(at + 2u * 0x9E3779B97F4A7C15ULL) >> 11) * 0x1.0p-53};
            v = 2.0 * (rv - D3{0.5, 0.5, 0.5});
            inside = dot(v, v) < 1.0;
        } else {
            // rand_scalar(-1, 1) = u * 2 + (-1) (exact either way: u has 53 bits); the test as the per-lane sampler's
            // code makes it, camera_ray
            v = D3{fma((double)(mix64(at) >> 11) * 0x1.0p-53, 2.0, -1.0), fma((double)(mix64(at + 0x9E3779B97F4A7C15ULL) >> 11) * 0x1.0p-53, 2.0, -1.0), 0.0};
            inside = dot(v, v) < 1.0;
        }
        const bool ok = q < n && attempt < MRT_MAX_REJECT && inside;
        const unsigned long long accepted = __ballot(ok);
        // (a lane without a request can have rank == n, i.e. a shift by 64: masked, its value is not used)
        const uint32_t mine = (uint32_t)(accepted >> ((rank << log_a) & 63u)) & ((1u << (1u << log_a)) - 1u);
        const bool found = want && mine != 0;
        const int sel = (int)((found ? (rank << log_a) + (uint32_t)__builtin_ctz(mine) : lane) << 2);
        D3 got;
        got.x = __hiloint2double(__builtin_amdgcn_ds_bpermute(sel, __double2hiint(v.x)), __builtin_amdgcn_ds_bpermute(sel, __double2loint(v.x)));
        got.y = __hiloint2double(__builtin_amdgcn_ds_bpermute(sel, __double2hiint(v.y)), __builtin_amdgcn_ds_bpermute(sel, __double2loint(v.y)));
        got.z = 0.0;
        if constexpr (DIMS == 3)
            got.z = __hiloint2double(__builtin_amdgcn_ds_bpermute(sel, __double2hiint(v.z)), __builtin_amdgcn_ds_bpermute(sel, __double2loint(v.z)));
        if (found) {
            out = got;
            want = false;
        } else if (want) {
            first += 1u << log_a;
            if (first >= MRT_MAX_REJECT) want = false;  // gives up with the zero vector, as rand_unit_vec does
        }
        mask = __ballot(want);
    }
    return out;
}
__device__ __forceinline__ D3 wave_rand_unit_vec(bool want, uint64_t rng_base, uint32_t dim_base) {
    return wave_rand_reject<3>(want, rng_base, dim_base);
}

// Scene records are read through two kinds of pointers: ordinary global pointers when the
// index differs per lane, and constant-address-space pointers (CONST_AS) when the index is the
// same for the whole wave, so that the loads become scalar loads and the values live in SGPRs.

// ============================================================== transforms (hittable.rs:92-107)
// The functions of the nearest-hit query below (transforms, shape and box tests) spell out their
// fused multiply-adds and switch the compiler's own contraction off: the same source then rounds the
// same way in every instantiation (scalar or per-lane operands, any feature set), which is what lets the
// traversal variants be compared bit for bit.
template <class AP>
__device__ __forceinline__ D3 apply(AP a, D3 p) {
#pragma clang fp contract(off)
    return {fma(a->r[0][2], p.z, fma(a->r[0][1], p.y, a->r[0][0] * p.x)) + a->r[0][3],
            fma(a->r[1][2], p.z, fma(a->r[1][1], p.y, a->r[1][0] * p.x)) + a->r[1][3],
            fma(a->r[2][2], p.z, fma(a->r[2][1], p.y, a->r[2][0] * p.x)) + a->r[2][3]};
}
// Transform::mul_ray: direction = M*(o+d) - M*o; never renormalised, so t is shared.
template <class AP>
__device__ __forceinline__ RayD xform_ray(AP a, bool identity, const RayD& r) {
#pragma clang fp contract(off)
    if (identity) return {r.o, D3{(r.o.x + r.d.x) - r.o.x, (r.o.y + r.d.y) - r.o.y, (r.o.z + r.d.z) - r.o.z}, r.time};
    D3 e = apply(a, D3{r.o.x + r.d.x, r.o.y + r.d.y, r.o.z + r.d.z});
    D3 o = apply(a, r.o);
    return {o, D3{e.x - o.x, e.y - o.y, e.z - o.z}, r.time};
}

// ============================================================== shape intersection
// rect.rs:63-86,185-206,309-331. AXIS = constant axis (2 XY, 1 XZ, 0 YZ). `inv` = 1/direction
// per component, shared by every plane test against the same ray.
template <int AXIS>
__device__ __forceinline__ bool rect_hit(double a0, double a1, double b0, double b1, double k, const RayD& r, D3 inv,
                                         double t_min, double t_max, double& t_out) {
#pragma clang fp contract(off)
    constexpr int IA = AXIS == 0 ? 1 : 0;
    constexpr int IB = AXIS == 2 ? 1 : 2;
    double t = (k - comp(r.o, AXIS)) * comp(inv, AXIS);
    if (t < t_min || t > t_max) return false;
    double a = fma(t, comp(r.d, IA), comp(r.o, IA));
    double b = fma(t, comp(r.d, IB), comp(r.o, IB));
    if (a < a0 || a > a1 || b < b0 || b > b1) return false;
    t_out = t;
    return true;
}
// sphere.rs:28-45 (also MovingSphere :145-162 with center(time))
__device__ __forceinline__ bool sphere_hit(D3 c, double radius, const RayD& r, double t_min, double t_max,
                                           double& t_out) {
#pragma clang fp contract(off)
    const double rx = r.o.x - c.x, ry = r.o.y - c.y, rz = r.o.z - c.z;
    double a = fma(r.d.z, r.d.z, fma(r.d.y, r.d.y, r.d.x * r.d.x));
    double half_b = fma(rz, r.d.z, fma(ry, r.d.y, rx * r.d.x));
    double cc = fma(-radius, radius, fma(rz, rz, fma(ry, ry, rx * rx)));
    double disc = fma(-a, cc, half_b * half_b);
    if (disc < 0.0) return false;
    double sq = sqrt(disc);
    double inv_a = rcp_fast(a);
    double root = (-half_b - sq) * inv_a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sq) * inv_a;
        if (root < t_min || t_max < root) return false;
    }
    t_out = root;
    return true;
}
// An upper bound (f32, rounded up) on the nearest hit of a ray that starts on an untransformed Sphere: the far root of
// sphere_hit's own quadratic for that sphere, evaluated with sphere_hit's expressions. If the root is a valid answer
// (>= t_min with room to spare), sphere_hit on this sphere returns it or the near root, whichever it accepts first,
// unless the running bound is below them already: either way the query ends at or below the far root -- provided the
// walk gets to test the sphere at all, i.e. the sphere's leaf box (centre -+ radius, the host's expressions) passes
// Aabb::hit for every running bound >= t_min. That is checked here per axis on the walk's own slab bounds: the
// entry (box face behind the ray - origin) * (1 / d) has a non-positive numerator, so it is <= 0 whatever the
// reciprocal's rounding, and the exit's numerator is at least 0.0011 d, so it is >= t_min. A negative radius (inside-out
// box) gives no bound. INFINITY when there is none.
// (Where it pays: the reference's Dielectric does not flip the normal for a ray that leaves the glass, its Schlick term
// exceeds 1 there and the ray always reflects -- material.rs:141-196 -- so a path that enters a glass sphere bounces
// inside it until the depth limit, and 71 % of the random-spheres scene's rays start on a sphere pointing into it.)
__device__ __forceinline__ float sphere_far_root_cap(D3 c, double radius, const RayD& r) {
#pragma clang fp contract(off)
    if (!(radius > 0.0)) return INFINITY;
    const double ox[3] = {r.o.x, r.o.y, r.o.z}, dx[3] = {r.d.x, r.d.y, r.d.z}, cx[3] = {c.x, c.y, c.z};
    bool box_ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double lo = (cx[i] - radius) - ox[i], hi = (cx[i] + radius) - ox[i];  // Aabb::hit's numerators
        const double behind = dx[i] > 0.0 ? lo : -hi, ahead = dx[i] > 0.0 ? hi : -lo;
        box_ok = box_ok && dx[i] != 0.0 && behind <= 0.0 && ahead >= 0.0011 * fabs(dx[i]);
    }
    if (!box_ok) return INFINITY;
    const double rx = r.o.x - c.x, ry = r.o.y - c.y, rz = r.o.z - c.z;
    double a = fma(r.d.z, r.d.z, fma(r.d.y, r.d.y, r.d.x * r.d.x));
    double half_b = fma(rz, r.d.z, fma(ry, r.d.y, rx * r.d.x));
    double cc = fma(-radius, radius, fma(rz, rz, fma(ry, ry, rx * rx)));
    double disc = fma(-a, cc, half_b * half_b);
    if (!(disc >= 0.0)) return INFINITY;
    const double far_root = (-half_b + sqrt(disc)) * rcp_fast(a);
    if (!(far_root >= 0.002 && far_root < 1e30)) return INFINITY;
    return (float)(far_root * 1.0000002);  // (more than half an f32 step: rounded up whatever the conversion does)
}
// The two boundary queries of a medium around a Sphere (constant_medium.rs:36-44): sphere_hit over
// [-1e10, 1e10], then over [t1 + 1e-4, 1e10]. Both answers are roots of the same quadratic, so the discriminant,
// its square root and the reciprocal are worked out once; the choice of the root is sphere_hit's, per query.
__device__ __forceinline__ bool sphere_hit_twice(D3 c, double radius, const RayD& r, double& t1, double& t2) {
#pragma clang fp contract(off)
    const double rx = r.o.x - c.x, ry = r.o.y - c.y, rz = r.o.z - c.z;
    double a = fma(r.d.z, r.d.z, fma(r.d.y, r.d.y, r.d.x * r.d.x));
    double half_b = fma(rz, r.d.z, fma(ry, r.d.y, rx * r.d.x));
    double cc = fma(-radius, radius, fma(rz, rz, fma(ry, ry, rx * rx)));
    double disc = fma(-a, cc, half_b * half_b);
    if (disc < 0.0) return false;
    double sq = sqrt(disc);
    double inv_a = rcp_fast(a);
    const double near_root = (-half_b - sq) * inv_a, far_root = (-half_b + sq) * inv_a;
    const double hi = 10000000000.0;
    double lo = -10000000000.0;
    double root = near_root;
    if (root < lo || hi < root) {
        root = far_root;
        if (root < lo || hi < root) return false;
    }
    t1 = root;
    lo = root + 0.0001;
    root = near_root;
    if (root < lo || hi < root) {
        root = far_root;
        if (root < lo || hi < root) return false;
    }
    t2 = root;
    return true;
}
// sphere.rs:139-142
template <class PP>
__device__ __forceinline__ D3 moving_center(PP p, double time) {
#pragma clang fp contract(off)
    double f = (time - p[7]) / (p[8] - p[7]);
    return D3{fma(f, p[4] - p[1], p[1]), fma(f, p[5] - p[2], p[2]), fma(f, p[6] - p[3], p[3])};
}
// render_box.rs:117-129: faces in order xy+, xy-, xz+, xz-, yz+, yz-; strictly smaller t replaces.
template <class PP>
__device__ __forceinline__ bool box_hit(PP p, const RayD& r, D3 inv, double t_min, double t_max, double& t_out,
                                        int& face) {
    bool any = false;
    double best = 0.0, t;
    int f = 0;
    if (rect_hit<2>(p[0], p[3], p[1], p[4], p[5], r, inv, t_min, t_max, t)) { any = true, best = t, f = 0; }
    if (rect_hit<2>(p[0], p[3], p[1], p[4], p[2], r, inv, t_min, t_max, t) && (!any || t < best)) { any = true, best = t, f = 1; }
    if (rect_hit<1>(p[0], p[3], p[2], p[5], p[4], r, inv, t_min, t_max, t) && (!any || t < best)) { any = true, best = t, f = 2; }
    if (rect_hit<1>(p[0], p[3], p[2], p[5], p[1], r, inv, t_min, t_max, t) && (!any || t < best)) { any = true, best = t, f = 3; }
    if (rect_hit<0>(p[1], p[4], p[2], p[5], p[3], r, inv, t_min, t_max, t) && (!any || t < best)) { any = true, best = t, f = 4; }
    if (rect_hit<0>(p[1], p[4], p[2], p[5], p[0], r, inv, t_min, t_max, t) && (!any || t < best)) { any = true, best = t, f = 5; }
    t_out = best;
    face = f;
    return any;
}

// The two boundary queries of a medium around a RenderBox (constant_medium.rs:36-44): nearest face
// hit in [-1e10, 1e10], then nearest in [t1 + 1e-4, 1e10]. Whether a plane's intersection point
// lies inside its face does not depend on the query range, so both queries share the six plane
// intersections; the selection keeps box_hit's order and "strictly smaller replaces" rule.
template <int AXIS>
__device__ __forceinline__ void rect_plane(double a0, double a1, double b0, double b1, double k, const RayD& r, D3 inv,
                                           double& t, bool& inside) {
#pragma clang fp contract(off)
    constexpr int IA = AXIS == 0 ? 1 : 0;
    constexpr int IB = AXIS == 2 ? 1 : 2;
    t = (k - comp(r.o, AXIS)) * comp(inv, AXIS);
    double a = fma(t, comp(r.d, IA), comp(r.o, IA));
    double b = fma(t, comp(r.d, IB), comp(r.o, IB));
    inside = !(a < a0 || a > a1 || b < b0 || b > b1);
}
template <class PP>
__device__ __forceinline__ bool box_hit_twice(PP p, const RayD& r, D3 inv, double& t1, double& t2) {
#pragma clang fp contract(off)
    double tf[6];
    bool in[6];
    rect_plane<2>(p[0], p[3], p[1], p[4], p[5], r, inv, tf[0], in[0]);
    rect_plane<2>(p[0], p[3], p[1], p[4], p[2], r, inv, tf[1], in[1]);
    rect_plane<1>(p[0], p[3], p[2], p[5], p[4], r, inv, tf[2], in[2]);
    rect_plane<1>(p[0], p[3], p[2], p[5], p[1], r, inv, tf[3], in[3]);
    rect_plane<0>(p[1], p[4], p[2], p[5], p[3], r, inv, tf[4], in[4]);
    rect_plane<0>(p[1], p[4], p[2], p[5], p[0], r, inv, tf[5], in[5]);
    double lo = -10000000000.0;
    const double hi = 10000000000.0;
    double best[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        bool any = false;
        double b = 0.0;
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            if (in[f] && !(tf[f] < lo || tf[f] > hi) && (!any || tf[f] < b)) any = true, b = tf[f];
        }
        if (!any) return false;
        best[q] = b;
        lo = b + 0.0001;
    }
    t1 = best[0], t2 = best[1];
    return true;
}

// Deterministic shapes (everything a ConstantMedium may wrap). `sub` = box face.
template <int FEAT, class PP>
__device__ __forceinline__ bool solid_hit_kp(int kind, PP p, const RayD& r, D3 inv, double t_min, double t_max,
                                             double& t, int& sub) {
    sub = 0;
    switch (kind) {
        case MRT_SHAPE_SPHERE:
            if constexpr ((FEAT & FEAT_SPHERE) != 0) return sphere_hit(D3{p[1], p[2], p[3]}, p[0], r, t_min, t_max, t);
            return false;
        case MRT_SHAPE_MOVING_SPHERE:
            if constexpr ((FEAT & FEAT_MOVING) != 0) return sphere_hit(moving_center(p, r.time), p[0], r, t_min, t_max, t);
            return false;
        case MRT_SHAPE_XY_RECT: return rect_hit<2>(p[0], p[1], p[2], p[3], p[4], r, inv, t_min, t_max, t);
        case MRT_SHAPE_XZ_RECT: return rect_hit<1>(p[0], p[1], p[2], p[3], p[4], r, inv, t_min, t_max, t);
        case MRT_SHAPE_YZ_RECT: return rect_hit<0>(p[0], p[1], p[2], p[3], p[4], r, inv, t_min, t_max, t);
        case MRT_SHAPE_RENDER_BOX: return box_hit(p, r, inv, t_min, t_max, t, sub);
        default: return false;
    }
}

// Aabb::hit (bvh.rs:18-33): every axis is tested on its own against the ORIGINAL [t_min, t_max].
// Straight-line: the three axes are independent, and a wave gains nothing from leaving after the first
// (some lane nearly always goes on), while every exit is a branch around a handful of instructions.
__device__ __forceinline__ bool aabb_hit(D3 mn, D3 mx, const RayD& r, D3 inv_d, double t_min, double t_max) {
#pragma clang fp contract(off)
    double x0 = (mn.x - r.o.x) * inv_d.x, x1 = (mx.x - r.o.x) * inv_d.x;
    double y0 = (mn.y - r.o.y) * inv_d.y, y1 = (mx.y - r.o.y) * inv_d.y;
    double z0 = (mn.z - r.o.z) * inv_d.z, z1 = (mx.z - r.o.z) * inv_d.z;
    if (inv_d.x < 0.0) { double s = x0; x0 = x1; x1 = s; }
    if (inv_d.y < 0.0) { double s = y0; y0 = y1; y1 = s; }
    if (inv_d.z < 0.0) { double s = z0; z0 = z1; z1 = s; }
    // a NaN slab bound (0 * inf) leaves t_min / t_max
    const bool fx = vmin_f64(x1, t_max) < vmax_f64(x0, t_min);
    const bool fy = vmin_f64(y1, t_max) < vmax_f64(y0, t_min);
    const bool fz = vmin_f64(z1, t_max) < vmax_f64(z0, t_min);
    return !(fx | fy | fz);
}
// ============================================================== voxel octree (fast_oct_tree/)
// Leaf / empty chunk that contains cell (x, y, z) (in range): get_chunk_size + get_chunk
// (ray_trace.rs:23-83) as one descent over the breadth-first node array; the child of a parent of
// size exponent s is picked by bit s - 1 of the cell (Node::world_pos_to_child_index,
// mod.rs:217-225: child order x*4 + y*2 + z).
// The walk moves from cell to neighbouring cell, so the parents on the path to the previous cell
// are remembered per lane (their child links, by size exponent, in LDS) and the descent restarts at
// the smallest parent that also contains the new cell: one or two levels up for most steps
// instead of the whole depth from the root.
constexpr int OCT_ANC_LEVELS = 16;
// The per-lane LDS arrays of this file (octree ancestor cache, candidate-search bits and stack) live
// in the workgroup's dynamic LDS, laid out [entry][thread], so that kernels can differ in workgroup
// size; the two users never meet in one kernel (see lds_bytes), both start at offset 0.
__device__ __forceinline__ uint32_t* dynamic_lds() {
    extern __shared__ uint32_t mrt_dynamic_lds[];
    return mrt_dynamic_lds;
}
#ifdef MRT_STAMPS
// diagnostic build: steps of the octree walk summed over lanes [0], step rounds summed over waves [1]
// (counted only with -DMRT_OCT_DIAG on top: two atomics per step round distort the cycle stamps)
__device__ unsigned long long g_oct_diag[6];  // [2] node loads, [3] lane lookups that loaded, [4] wave lookups in which a lane loaded, [5] longest load chains summed over those
#endif
struct OctCursor {
    uint2* anc;         // this lane's column of a [OCT_ANC_LEVELS][threads per workgroup] array: (child link, child kinds) of a parent
    uint32_t stride;    // threads per workgroup
};
// What a lookup returns: the leaf or empty node that contains the cell -- its kind (MRT_OCT_EMPTY, MRT_OCT_LEAF, or
// OCT_LEAF_VOLUME for a leaf whose voxel is a volume), its size exponent and its index in the node array. The
// voxel record index of a leaf is that node's link, read only by the caller that needs it (the end of a walk).
constexpr uint32_t OCT_LEAF_VOLUME = 3;
struct OctCell {
    uint32_t kind, size, node;
};
// One node, by a load that names the global address space (a node array whose address went through a register
// pin has lost that knowledge, and a generic-space load counts against both memory counters).
__device__ __forceinline__ DOctNode oct_node_at(const DOctNode* nodes, uint32_t i) {
    const uint64_t raw = ((const __attribute__((address_space(1))) uint64_t*)nodes)[i];
    return DOctNode{(uint32_t)raw, (int32_t)(raw >> 32)};
}
// A parent's record carries the kinds of its eight children (two bits each, DOctNode::kind_size bits 16..31), so
// the descent reads a node only to enter a child that is itself a parent: a step to a sibling leaf or empty cell
// -- half of all steps -- is answered from the remembered parent alone, with no access to memory, and the chain
// of dependent loads per step, which bounds the walk, is shorter by one everywhere else.
// The walk's lookup: the lane's remembered path holds the parents of the node that the previous lookup returned
// (oct_cursor_start put the root there), which the caller names by any cell in it -- the walk's current block
// (px, py, pz) -- and its size exponent `pexp`; the descent starts at the smallest remembered parent that
// contains both that node and the new cell.
__device__ __forceinline__ OctCell oct_chunk_near(const DOctNode* nodes, const OctCursor& c, int32_t px, int32_t py, int32_t pz,
                                                  uint32_t pexp, int32_t x, int32_t y, int32_t z) {
    const uint32_t diff = (uint32_t)((x ^ px) | (y ^ py) | (z ^ pz));
    const uint32_t h = 32u - (uint32_t)__clz((int)diff);  // cells agree above bit h - 1 (__clz(0) = 32)
    uint32_t s = h > pexp + 1 ? h : pexp + 1;             // smallest remembered parent that holds both
    const uint2 e = c.anc[s * c.stride];
    uint32_t link = e.x, kinds = e.y, ci, k;
#ifdef MRT_OCT_DIAG
    uint32_t n_loads = 0;
#endif
    for (;;) {
        s -= 1;
        // (bit s of each coordinate by the bit-field instruction: five instructions for the eight of shifts and masks)
        ci = (__builtin_amdgcn_ubfe((uint32_t)x, s, 1u) << 2) | (__builtin_amdgcn_ubfe((uint32_t)y, s, 1u) << 1) |
             __builtin_amdgcn_ubfe((uint32_t)z, s, 1u);
        k = (kinds >> (2u * ci)) & 3u;
        if (k != MRT_OCT_PARENT) break;
        const DOctNode n = oct_node_at(nodes, link + ci);
        link = (uint32_t)n.link, kinds = n.kind_size >> 16;
        c.anc[s * c.stride] = uint2{link, kinds};
#ifdef MRT_OCT_DIAG
        n_loads++;
#endif
    }
#ifdef MRT_OCT_DIAG
    {
        const unsigned long long act = __ballot(1), ld = __ballot(n_loads > 0);
        uint32_t mx = n_loads;
        for (int o = 32; o > 0; o >>= 1) { uint32_t other = (uint32_t)__shfl_xor((int)mx, o, 64); mx = other > mx ? other : mx; }
        uint32_t total = n_loads;
        for (int o = 32; o > 0; o >>= 1) total += (uint32_t)__shfl_xor((int)total, o, 64);
        if ((int)__lane_id() == __ffsll((long long)act) - 1) {
            atomicAdd(&g_oct_diag[2], (unsigned long long)total), atomicAdd(&g_oct_diag[3], (unsigned long long)__popcll(ld));
            if (ld) atomicAdd(&g_oct_diag[4], 1ull), atomicAdd(&g_oct_diag[5], (unsigned long long)mx);
        }
    }
#endif
    return OctCell{k, s, link + ci};
}
// The same lookup from the root, without a remembered path: trees whose root is not a parent (one node) and
// worlds of 2^OCT_ANC_LEVELS cells or more along an edge, which the cursor's array does not hold.
__device__ OctCell oct_chunk_from_root(const DOctNode* nodes, uint32_t root, uint32_t world_log2, int32_t x, int32_t y, int32_t z) {
    DOctNode n = oct_node_at(nodes, root);
    if ((n.kind_size & 0xFF) != MRT_OCT_PARENT) return OctCell{(n.kind_size >> 16) & 3u, world_log2, root};
    uint32_t s = world_log2;
    for (;;) {
        s -= 1;
        const uint32_t ci = ((((uint32_t)x >> s) & 1u) << 2) | ((((uint32_t)y >> s) & 1u) << 1) | (((uint32_t)z >> s) & 1u);
        const uint32_t k = ((n.kind_size >> 16) >> (2u * ci)) & 3u;
        const uint32_t at = (uint32_t)n.link + ci;
        if (k != MRT_OCT_PARENT) return OctCell{k, s, at};
        n = oct_node_at(nodes, at);
    }
}
// Puts the root on the lane's remembered path, so that the first lookup of a walk (previous node: the whole world,
// exponent world_log2 - 1 at cell 0) descends from it like any other from its smallest common parent.
__device__ __forceinline__ void oct_cursor_start(const DOctNode* nodes, const DOctree& oc, const OctCursor& c) {
    const DOctNode n = oct_node_at(nodes, oc.root);
    c.anc[oc.world_log2 * c.stride] = uint2{(uint32_t)n.link, n.kind_size >> 16};
}
// `CACHED` (the same for every lane that walks this tree, DOctree::cached): the tree has a parent root and fits
// the remembered path.
template <bool CACHED>
__device__ __forceinline__ OctCell oct_lookup(const DOctNode* nodes, const DOctree& oc, const OctCursor& c, int32_t px, int32_t py,
                                              int32_t pz, uint32_t pexp, int32_t x, int32_t y, int32_t z) {
    if constexpr (CACHED) return oct_chunk_near(nodes, c, px, py, pz, pexp, x, y, z);
    else return oct_chunk_from_root(nodes, oc.root, oc.world_log2, x, y, z);
}
// Rust `as i32`: toward zero, saturating, NaN -> 0 -- which is what v_cvt_i32_f64 does by itself
// (a C++ cast leaves the out-of-range cases undefined, hence the instruction by name).
__device__ __forceinline__ int32_t f64_as_i32(double v) {
    int32_t r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ int32_t floor_size(int32_t v, uint32_t s) { return (v >> s) << s; }  // ray_trace.rs:686-688

// What an octree walk returns, packed into HitD::sub as voxel * OCT_SUB + code:
//   0..5   solid hit, the voxel's own material, normal code 0 -x, 1 +x, 2 -y, 3 +y, 4 -z, 5 +z
//   6..11  solid hit by the edge effect of a volume voxel (its solid_material), normal code - 6
//   12     volume hit: the free path ended inside the volume (material Volume { color }, normal +x)
constexpr int OCT_SUB = 16;
constexpr int OCT_EDGE = 6, OCT_VOLUME = 12;

// RayTraceState::volume_distance_left (ray_trace.rs:93-104)
struct OctVolume {
    bool some;
    double distance_left, first_density;
    D3 last_position;
    int previous;  // previous_volume: voxel record index
};
enum : int { OCT_CONTINUE = 0, OCT_STOP = 1 };
__device__ __forceinline__ D3 oct_stop_position(const OctVolume& vs, D3 d) {  // last + left * d.normalize()
    double inv = 1.0 / sqrt(dot(d, d));
    return vs.last_position + (d * inv) * vs.distance_left;
}
// handle_volume, ray_trace.rs:118-179. On OCT_STOP, `code` / `voxel` / `hit_pos` describe the hit.
__device__ int oct_handle_volume(const mrt_voxel* voxels, int v, OctVolume& vs, D3 cur, D3 d, int normal_code, const Rng& g,
                                 uint32_t& entry, D3& hit_pos, int& code, int& voxel) {
    if (vs.some) {
        double density_ratio = vs.first_density / voxels[vs.previous].density;
        D3 dl = cur - vs.last_position;
        double distance_traveled = sqrt(dot(dl, dl)) * density_ratio;
        if (distance_traveled > vs.distance_left) {
            hit_pos = oct_stop_position(vs, d);
            code = OCT_VOLUME, voxel = vs.previous;
            return OCT_STOP;
        }
        vs.distance_left = vs.distance_left - distance_traveled;
        vs.last_position = cur;
        return OCT_CONTINUE;
    }
    const mrt_voxel& vm = voxels[v];
    const uint32_t n = entry++;
    if (vm.edge_kind != MRT_VOX_EDGE_NONE) {
        float random_number = (float)rand_scalar(g, 0.0, 1.0, MRT_DIM_OCT_VOLUME + 2 * n);
        if (random_number < vm.edge_probability) {
            hit_pos = cur;
            code = OCT_EDGE + normal_code, voxel = v;
            return OCT_STOP;
        }
    }
    vs.some = true;
    vs.distance_left = log(rand_scalar(g, 0.0, 1.0, MRT_DIM_OCT_VOLUME + 2 * n + 1)) / (-vm.density);
    vs.last_position = cur;
    vs.first_density = vm.density;
    vs.previous = v;
    return OCT_CONTINUE;
}
// handle_empty, ray_trace.rs:180-199
__device__ __forceinline__ int oct_handle_empty(OctVolume& vs, D3 cur, D3 d, D3& hit_pos, int& code, int& voxel) {
    if (!vs.some) return OCT_CONTINUE;
    D3 dl = cur - vs.last_position;
    if (sqrt(dot(dl, dl)) > vs.distance_left) {
        hit_pos = oct_stop_position(vs, d);
        code = OCT_VOLUME, voxel = vs.previous;
        return OCT_STOP;
    }
    vs.some = false;
    return OCT_CONTINUE;
}

// ray_iteration (ray_trace.rs:86-560): walk leaf to leaf until a solid leaf is entered, or (VOL) a
// volume's edge effect or free path stops the ray. `cur` starts at the ray origin. Returns false
// for None. `code`: see OCT_SUB.
template <bool VOL, bool CACHED>
__device__ bool oct_ray_iteration(const DOctNode* nodes, const mrt_voxel* voxels, const DOctree& oc, int32_t bx, int32_t by,
                                  int32_t bz, D3 origin, D3 d, int initial_normal, const Rng& g, uint32_t& entry, D3& hit_pos,
                                  int& code, int& voxel, const OctCursor& cursor, int32_t qx, int32_t qy, int32_t qz, uint32_t qexp) {
    const uint32_t ws = 1u << oc.world_log2;
    // The node array's address is a field of the scene view in constant memory, which the compiler would rather
    // read again at every lookup than keep: a scalar round trip in front of each step's node load. As an opaque
    // value it stays in registers for the walk.
    asm volatile("" : "+s"(nodes));
    auto in_range = [&](int32_t x, int32_t y, int32_t z) {  // 0 <= v < ws, negative values wrap above ws
        return (uint32_t)x < ws && (uint32_t)y < ws && (uint32_t)z < ws;
    };
    if (!in_range(bx, by, bz)) return false;  // get_chunk(..).expect("out of range")
    // (qx, qy, qz, qexp: the node of the lookup before this walk, see oct_chunk_near)
    const OctCell chunk = oct_lookup<CACHED>(nodes, oc, cursor, qx, qy, qz, qexp, bx, by, bz);
    uint32_t csz = chunk.size;
    bx = floor_size(bx, csz), by = floor_size(by, csz), bz = floor_size(bz, csz);
    D3 cur = origin;
    OctVolume vs{false, 0.0, 0.0, {0.0, 0.0, 0.0}, -1};
    // the starting cell: a volume starts the free-path walk, a solid is ignored (ray_trace.rs:476-510)
    if constexpr (VOL) {
        if (chunk.kind == OCT_LEAF_VOLUME) {
            if (oct_handle_volume(voxels, oct_node_at(nodes, chunk.node).link, vs, cur, d, initial_normal, g, entry, hit_pos, code, voxel) == OCT_STOP)
                return true;
        }
    }
    // the far face of the block on each axis: block + step for a non-negative direction component
    // (ray_trace.rs adds step * 1.0 or step * 0.0 in floating point; same integer either way)
    const bool px = !__builtin_signbit(d.x), py = !__builtin_signbit(d.y), pz = !__builtin_signbit(d.z);
    const bool fx = d.x >= 0.0, fy = d.y >= 0.0, fz = d.z >= 0.0;
    const D3 inv{rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z)};
    int32_t step = 1 << csz;  // get_step_size of the current block
    // A solid hit leaves the loop and is written out after it: assigned inside, `hit_pos` would be a value carried
    // around the loop, three 64-bit copies per step.
    uint32_t solid_node = ~0u;
    int solid_code = 0;
    for (int it = 0; it < 3000; ++it) {
#ifdef MRT_OCT_DIAG
        {
            const unsigned long long act = __ballot(1);
            if ((int)__lane_id() == __ffsll((long long)act) - 1)
                atomicAdd(&g_oct_diag[0], (unsigned long long)__popcll(act)), atomicAdd(&g_oct_diag[1], 1ull);
        }
#endif
        // (the block is in range here: checked on entry, and every step below ends in a checked cell's block)
        double tx = ((double)(bx + (px ? step : 0)) - cur.x) * inv.x;
        double ty = ((double)(by + (py ? step : 0)) - cur.y) * inv.y;
        double tz = ((double)(bz + (pz ? step : 0)) - cur.z) * inv.z;
        if (tx < 0.0 || ty < 0.0 || tz < 0.0) break;
        // The reference has six copies of this step (axis x sign, ray_trace.rs:222-470); here the axis
        // is a value, so the lanes of a wave take one path and issue one lookup together. The
        // smallest t picks the axis, ties fall to z (ray_trace.rs:398).
        // (bitwise, not short-circuit: the compiler turns `&&` between comparisons into branches here)
        const bool ax0 = (tx < ty) & (tx < tz), ax1 = !ax0 & (ty < tx) & (ty < tz), ax2 = !ax0 & !ax1;
        const double txy = vmin_f64(tx, ty);
        const double tm = ax2 ? tz : txy;  // x or y win only as the strict minimum
        const int32_t ba = ax0 ? bx : (ax1 ? by : bz);
        cur.x += tm * d.x, cur.y += tm * d.y, cur.z += tm * d.z;
        const int32_t cx = f64_as_i32(cur.x), cy = f64_as_i32(cur.y), cz = f64_as_i32(cur.z);
        // forwards (component >= 0, which includes -0.0): the block coordinate moves by the step, whose
        // sign follows the sign bit; backwards: the cell just below the block
        const bool forward = (ax0 && fx) || (ax1 && fy) || (ax2 && fz);
        const bool plus = (ax0 && px) || (ax1 && py) || (ax2 && pz);
        const int32_t la = forward ? ba + (plus ? step : -step) : ba - 1;
        const int32_t lx = ax0 ? la : cx, ly = ax1 ? la : cy, lz = ax2 ? la : cz;
        // get_chunk_size panics out of range; this also covers the reference's earlier check of the
        // cell below the old block on the way back (the other two coordinates are the block's own)
        if (!in_range(lx, ly, lz)) break;
        const OctCell node = oct_lookup<CACHED>(nodes, oc, cursor, bx, by, bz, csz, lx, ly, lz);
        const uint32_t ncs = node.size;
        const int32_t na = forward ? la : ba - (1 << ncs);
        bx = ax0 ? na : floor_size(cx, ncs);
        by = ax1 ? na : floor_size(cy, ncs);
        bz = ax2 ? na : floor_size(cz, ncs);
        const int normal_code = (ax0 ? 0 : (ax1 ? 2 : 4)) + (forward ? 0 : 1);
        // handle_hit, ray_trace.rs:202-221
        if (node.kind != MRT_OCT_EMPTY) {
            const bool is_volume = VOL && node.kind == OCT_LEAF_VOLUME;
            if (!is_volume) {  // StopIterationSolid with the voxel's own material
                solid_node = node.node, solid_code = normal_code;
                break;
            }
            if constexpr (VOL) {
                if (oct_handle_volume(voxels, oct_node_at(nodes, node.node).link, vs, cur, d, normal_code, g, entry, hit_pos, code, voxel) == OCT_STOP)
                    return true;
            }
        } else if constexpr (VOL) {
            if (oct_handle_empty(vs, cur, d, hit_pos, code, voxel) == OCT_STOP) return true;
        }
        step = 1 << ncs, csz = ncs;
    }
    if (solid_node != ~0u) {
        hit_pos = cur;
        code = solid_code, voxel = oct_node_at(nodes, solid_node).link;
        return true;
    }
    return false;  // "Max number of iterations reached"
}

// FastOctTree::trace_ray (ray_trace.rs:561-683); `d` is the normalised direction.
template <bool VOL, bool CACHED>
__device__ bool oct_trace_ray(const DOctNode* nodes, const mrt_voxel* voxels, const DOctree& oc, D3 o, D3 d, const Rng& g,
                              uint32_t& entry, D3& hit_pos, int& code, int& voxel) {
    const double ws = (double)(1u << oc.world_log2);
    // (the workgroup's dynamic LDS: kernels with octree code are launched with lds_bytes() of it)
    const OctCursor cursor{reinterpret_cast<uint2*>(dynamic_lds()) + threadIdx.x, blockDim.x};
    if constexpr (CACHED) oct_cursor_start(nodes, oc, cursor);
    int32_t px = 0, py = 0, pz = 0;         // the node of the previous lookup: a cell in it and its size exponent
    uint32_t pexp = oc.world_log2 - 1;
    // where the walk starts: at the origin when it lies in the world's cube, else where the ray enters the cube
    D3 position = o;
    int32_t bx = f64_as_i32(o.x), by = f64_as_i32(o.y), bz = f64_as_i32(o.z);
    int initial_normal = 1;  // unit_x
    if (!(o.x >= 0.0 && o.x <= ws && o.y >= 0.0 && o.y <= ws && o.z >= 0.0 && o.z <= ws)) {
        // (the six quotients of ray_trace.rs:575-600 by the direction's reciprocals, div_by_rcp)
        const D3 id{rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z)};
        double v0 = div_by_rcp(0.0 - o.x, d.x, id.x), v1 = div_by_rcp(ws - o.x, d.x, id.x);
        double v2 = div_by_rcp(0.0 - o.y, d.y, id.y), v3 = div_by_rcp(ws - o.y, d.y, id.y);
        double v4 = div_by_rcp(0.0 - o.z, d.z, id.z), v5 = div_by_rcp(ws - o.z, d.z, id.z);
        // entry = the largest of the per-axis near planes, exit = the smallest of the far planes
        int min_x = v0 < v1 ? 0 : 1, min_y = v2 < v3 ? 2 : 3, min_z = v4 < v5 ? 4 : 5;
        double nx = v0 < v1 ? v0 : v1, fx = v0 < v1 ? v1 : v0;
        double ny = v2 < v3 ? v2 : v3, fy = v2 < v3 ? v3 : v2;
        double nz = v4 < v5 ? v4 : v5, fz = v4 < v5 ? v5 : v4;
        int min_index;
        double vmin;
        if (nx > ny && nx > nz) min_index = min_x, vmin = nx;
        else if (ny > nx && ny > nz) min_index = min_y, vmin = ny;
        else min_index = min_z, vmin = nz;
        double vmax;
        if (fx < fy && fx < fz) vmax = fx;
        else if (fy < fx && fy < fz) vmax = fy;
        else vmax = fz;
        if (!(vmax >= vmin && vmin >= 0.0)) return false;
        position = o + vmin * d;
        // BLOCK_OFFSETS: -1 on the axis of a "far side" entry plane (indices 1, 3, 5)
        bx = f64_as_i32(position.x) + (min_index == 1 ? -1 : 0);
        by = f64_as_i32(position.y) + (min_index == 3 ? -1 : 0);
        bz = f64_as_i32(position.z) + (min_index == 5 ? -1 : 0);
        initial_normal = min_index;
        const int32_t wsi = (int32_t)(1u << oc.world_log2);
        // self.get(block as u32): a negative coordinate wraps and is out of range -> None
        if (bx >= 0 && bx < wsi && by >= 0 && by < wsi && bz >= 0 && bz < wsi) {
            const OctCell n = oct_lookup<CACHED>(nodes, oc, cursor, px, py, pz, pexp, bx, by, bz);
            px = bx, py = by, pz = bz, pexp = n.size;
            if (n.kind != MRT_OCT_EMPTY && !(VOL && n.kind == OCT_LEAF_VOLUME)) {  // a solid entry voxel is the hit (ray_trace.rs:653-657)
                hit_pos = position;
                code = min_index;  // NORMALS[min_index]: 0 -x, 1 +x, 2 -y, 3 +y, 4 -z, 5 +z
                voxel = oct_node_at(nodes, n.node).link;
                return true;
            }
        }
    }
    return oct_ray_iteration<VOL, CACHED>(nodes, voxels, oc, bx, by, bz, position, d, initial_normal, g, entry, hit_pos, code, voxel, cursor,
                                          px, py, pz, pexp);
}

// Hittable for FastOctTree<Voxel> (fast_oct_tree/hittable.rs:10-68); `sr` is the object-space ray.
template <bool VOL>
__device__ bool octree_hit(const DOctNode* nodes, const mrt_voxel* voxels, const DOctree& oc, const RayD& sr, D3 inv,
                           double t_min, double t_max, const Rng& g, uint32_t& entry, double& t_out, int& sub, D3& pos_out) {
    if (oc.n_nodes == 0) return false;
    const double ws = (double)(1u << oc.world_log2);
    if (!aabb_hit(D3{0.0, 0.0, 0.0}, D3{ws, ws, ws}, sr, inv, t_min, t_max)) return false;
    D3 nd = normalize(sr.d);
    D3 hp;
    int code, voxel;
    // (`cached`, worked out by the host: the tree has a parent root and fits the walk's remembered path)
    if (oc.cached ? !oct_trace_ray<VOL, true>(nodes, voxels, oc, sr.o, nd, g, entry, hp, code, voxel)
                  : !oct_trace_ray<VOL, false>(nodes, voxels, oc, sr.o, nd, g, entry, hp, code, voxel))
        return false;
    D3 dp = hp - sr.o;
    double t = sqrt(dot(dp, dp) / dot(sr.d, sr.d));
    if (!(t > t_min && t < t_max)) return false;  // strict, unlike the other shapes
    t_out = t;
    sub = voxel * OCT_SUB + code;
    pos_out = hp;
    return true;
}

// ---- a walk that can be put down and taken up again (solid-voxel trees with a remembered path)
// In a wave, walks differ in length: in the voxel city a ray takes 12 steps on average and its wave waited 21 rounds for
// the longest, 36 of 64 lanes stepping. So a walk can stop at a step boundary: when fewer than OCT_SUSPEND_BELOW lanes are
// still walking (and the others have something to do) those lanes keep their walk -- position, block, its size and
// the step count in LDS, the remembered path where it is -- while the others shade, start new paths and join them in the
// next turn of the kernel's loop. The steps a ray takes are the same, one after the other, whatever turn they are taken in.
struct OctWalk {
    D3 cur;              // position (ray_iteration's current_position)
    int32_t bx, by, bz;  // current block
    uint32_t csz;        // its size exponent
    int32_t it;          // steps taken (the reference stops at 3000)
};
constexpr int OCT_WALK_DWORDS = 11;
#ifndef MRT_OCT_T
#define MRT_OCT_T 8
#endif
#ifndef MRT_OCT_RMIN
#define MRT_OCT_RMIN 4
#endif
constexpr int OCT_SUSPEND_BELOW = MRT_OCT_T, OCT_SUSPEND_MIN_ROUNDS = MRT_OCT_RMIN;
enum : int { OCT_WALKING = 0, OCT_MISS = 1, OCT_SOLID = 2, OCT_IDLE = 3 };
__device__ __forceinline__ uint32_t* oct_walk_lds() {  // this lane's column, behind the remembered paths
    return dynamic_lds() + 2 * OCT_ANC_LEVELS * blockDim.x + threadIdx.x;
}
__device__ __forceinline__ void oct_walk_save(const OctWalk& w) {
    uint32_t* q = oct_walk_lds();
    const uint32_t st = blockDim.x;
    q[0 * st] = (uint32_t)__double2loint(w.cur.x), q[1 * st] = (uint32_t)__double2hiint(w.cur.x);
    q[2 * st] = (uint32_t)__double2loint(w.cur.y), q[3 * st] = (uint32_t)__double2hiint(w.cur.y);
    q[4 * st] = (uint32_t)__double2loint(w.cur.z), q[5 * st] = (uint32_t)__double2hiint(w.cur.z);
    q[6 * st] = (uint32_t)w.bx, q[7 * st] = (uint32_t)w.by, q[8 * st] = (uint32_t)w.bz, q[9 * st] = w.csz, q[10 * st] = (uint32_t)w.it;
}
__device__ __forceinline__ OctWalk oct_walk_load() {
    const uint32_t* q = oct_walk_lds();
    const uint32_t st = blockDim.x;
    OctWalk w;
    w.cur.x = __hiloint2double((int)q[1 * st], (int)q[0 * st]);
    w.cur.y = __hiloint2double((int)q[3 * st], (int)q[2 * st]);
    w.cur.z = __hiloint2double((int)q[5 * st], (int)q[4 * st]);
    w.bx = (int32_t)q[6 * st], w.by = (int32_t)q[7 * st], w.bz = (int32_t)q[8 * st], w.csz = q[9 * st], w.it = (int32_t)q[10 * st];
    return w;
}
// The direction's constants of a walk, worked out where a walk starts or is taken up again.
struct OctDir {
    D3 d, inv;
    bool px, py, pz, fx, fy, fz;  // sign bit clear / component >= 0 (they differ for -0.0)
};
__device__ __forceinline__ OctDir oct_dir(D3 d) {
    return OctDir{d, D3{rcp_fast(d.x), rcp_fast(d.y), rcp_fast(d.z)}, !__builtin_signbit(d.x), !__builtin_signbit(d.y), !__builtin_signbit(d.z),
                  d.x >= 0.0, d.y >= 0.0, d.z >= 0.0};
}
// One step of ray_iteration (ray_trace.rs:200-470) in a tree of solid voxels: the body of oct_ray_iteration's loop.
// OCT_SOLID: the step entered a solid leaf, node `solid_node`, normal code `solid_code`, position w.cur.
__device__ __forceinline__ int oct_step_solid(const DOctNode* nodes, const OctCursor& cursor, uint32_t ws, const OctDir& dr, OctWalk& w,
                                              uint32_t& solid_node, int& solid_code) {
    if (w.it >= 3000) return OCT_MISS;  // "Max number of iterations reached"
    w.it++;
    const int32_t step = 1 << w.csz;
    const double tx = ((double)(w.bx + (dr.px ? step : 0)) - w.cur.x) * dr.inv.x;
    const double ty = ((double)(w.by + (dr.py ? step : 0)) - w.cur.y) * dr.inv.y;
    const double tz = ((double)(w.bz + (dr.pz ? step : 0)) - w.cur.z) * dr.inv.z;
    if (tx < 0.0 || ty < 0.0 || tz < 0.0) return OCT_MISS;
    const bool ax0 = (tx < ty) & (tx < tz), ax1 = !ax0 & (ty < tx) & (ty < tz), ax2 = !ax0 & !ax1;
    const double txy = vmin_f64(tx, ty);
    const double tm = ax2 ? tz : txy;
    const int32_t ba = ax0 ? w.bx : (ax1 ? w.by : w.bz);
    w.cur.x += tm * dr.d.x, w.cur.y += tm * dr.d.y, w.cur.z += tm * dr.d.z;
    const int32_t cx = f64_as_i32(w.cur.x), cy = f64_as_i32(w.cur.y), cz = f64_as_i32(w.cur.z);
    const bool forward = (ax0 && dr.fx) || (ax1 && dr.fy) || (ax2 && dr.fz);
    const bool plus = (ax0 && dr.px) || (ax1 && dr.py) || (ax2 && dr.pz);
    const int32_t la = forward ? ba + (plus ? step : -step) : ba - 1;
    const int32_t lx = ax0 ? la : cx, ly = ax1 ? la : cy, lz = ax2 ? la : cz;
    if (!((uint32_t)lx < ws && (uint32_t)ly < ws && (uint32_t)lz < ws)) return OCT_MISS;
    const OctCell node = oct_chunk_near(nodes, cursor, w.bx, w.by, w.bz, w.csz, lx, ly, lz);
    const uint32_t ncs = node.size;
    const int32_t na = forward ? la : ba - (1 << ncs);
    w.bx = ax0 ? na : floor_size(cx, ncs);
    w.by = ax1 ? na : floor_size(cy, ncs);
    w.bz = ax2 ? na : floor_size(cz, ncs);
    w.csz = ncs;
    if (node.kind != MRT_OCT_EMPTY) {
        solid_node = node.node;
        solid_code = (ax0 ? 0 : (ax1 ? 2 : 4)) + (forward ? 0 : 1);
        return OCT_SOLID;
    }
    return OCT_WALKING;
}
// trace_ray up to the first step (ray_trace.rs:561-683 and the head of ray_iteration, :472-510), solid voxels.
__device__ __forceinline__ int oct_walk_begin(const DOctNode* nodes, const DOctree& oc, const OctCursor& cursor, D3 o, const OctDir& dr, OctWalk& w,
                                              uint32_t& solid_node, int& solid_code) {
    const D3 d = dr.d;
    const double ws = (double)(1u << oc.world_log2);
    const uint32_t wsu = 1u << oc.world_log2;
    D3 position = o;
    int32_t bx = f64_as_i32(o.x), by = f64_as_i32(o.y), bz = f64_as_i32(o.z);
    int32_t qx = 0, qy = 0, qz = 0;
    uint32_t qexp = oc.world_log2 - 1;
    if (!(o.x >= 0.0 && o.x <= ws && o.y >= 0.0 && o.y <= ws && o.z >= 0.0 && o.z <= ws)) {
        // (the six quotients of ray_trace.rs:575-600 by the direction's reciprocals, div_by_rcp)
        double v0 = div_by_rcp(0.0 - o.x, d.x, dr.inv.x), v1 = div_by_rcp(ws - o.x, d.x, dr.inv.x);
        double v2 = div_by_rcp(0.0 - o.y, d.y, dr.inv.y), v3 = div_by_rcp(ws - o.y, d.y, dr.inv.y);
        double v4 = div_by_rcp(0.0 - o.z, d.z, dr.inv.z), v5 = div_by_rcp(ws - o.z, d.z, dr.inv.z);
        int min_x = v0 < v1 ? 0 : 1, min_y = v2 < v3 ? 2 : 3, min_z = v4 < v5 ? 4 : 5;
        double nx = v0 < v1 ? v0 : v1, fx = v0 < v1 ? v1 : v0;
        double ny = v2 < v3 ? v2 : v3, fy = v2 < v3 ? v3 : v2;
        double nz = v4 < v5 ? v4 : v5, fz = v4 < v5 ? v5 : v4;
        int min_index;
        double vmin;
        if (nx > ny && nx > nz) min_index = min_x, vmin = nx;
        else if (ny > nx && ny > nz) min_index = min_y, vmin = ny;
        else min_index = min_z, vmin = nz;
        double vmax;
        if (fx < fy && fx < fz) vmax = fx;
        else if (fy < fx && fy < fz) vmax = fy;
        else vmax = fz;
        if (!(vmax >= vmin && vmin >= 0.0)) return OCT_MISS;
        position = o + vmin * d;
        bx = f64_as_i32(position.x) + (min_index == 1 ? -1 : 0);
        by = f64_as_i32(position.y) + (min_index == 3 ? -1 : 0);
        bz = f64_as_i32(position.z) + (min_index == 5 ? -1 : 0);
        if ((uint32_t)bx < wsu && (uint32_t)by < wsu && (uint32_t)bz < wsu) {
            const OctCell n = oct_chunk_near(nodes, cursor, qx, qy, qz, qexp, bx, by, bz);
            qx = bx, qy = by, qz = bz, qexp = n.size;
            if (n.kind != MRT_OCT_EMPTY) {  // a solid entry voxel is the hit (ray_trace.rs:653-657)
                w.cur = position;
                solid_node = n.node, solid_code = min_index;
                return OCT_SOLID;
            }
        }
    }
    if (!((uint32_t)bx < wsu && (uint32_t)by < wsu && (uint32_t)bz < wsu)) return OCT_MISS;  // get_chunk(..).expect("out of range")
    const OctCell chunk = oct_chunk_near(nodes, cursor, qx, qy, qz, qexp, bx, by, bz);
    w.cur = position;
    w.csz = chunk.size;
    w.bx = floor_size(bx, w.csz), w.by = floor_size(by, w.csz), w.bz = floor_size(bz, w.csz);
    w.it = 0;
    return OCT_WALKING;  // (a solid starting cell is ignored, ray_trace.rs:476-510)
}
// The octree leaf that the wave-synchronous walk visits first (SceneView::oct_first), as Object::hit + Hittable for
// FastOctTree (hittable.rs:204-237, fast_oct_tree/hittable.rs:10-68) with a walk that can be suspended. `fresh`: lanes
// that come to the leaf now and passed its box; `resuming`: lanes whose walk was suspended in the kernel loop's previous
// turn. On return `suspended` marks the lanes whose walk is put down again; the others have their answer in
// closest / best_*. Every lane of the wave must call this.
template <class RP>
__device__ __forceinline__ void octree_first_leaf(SceneRef sv, RP r, int flags, const RayD& ray, D3 inv_world, bool fresh, bool resuming,
                                                  bool may_suspend, double t_min, double& closest, int& best_obj, int& best_sub,
                                                  D3& best_pos, bool& suspended) {
#pragma clang fp contract(off)
    const DOctNode* nodes = sv.oct_nodes;
    asm volatile("" : "+s"(nodes));
    const DOctree& oc = sv.octrees[r->octree];
    RayD sr = ray;
    D3 inv = inv_world;
    if ((flags & NODE_IDENTITY) == 0) {
        if ((flags & NODE_TRANSLATION) != 0) {
            sr.o = {ray.o.x + r->to_object.r[0][3], ray.o.y + r->to_object.r[1][3], ray.o.z + r->to_object.r[2][3]};
        } else {
            sr = xform_ray(&r->to_object, false, ray);
            inv = {rcp_fast(sr.d.x), rcp_fast(sr.d.y), rcp_fast(sr.d.z)};
        }
    }
    // (only the segment kernels of the wave-synchronous walk come here, single-wave workgroups: the stride of the
    // per-lane LDS arrays is a constant, and a level's address a shift instead of a multiplication)
    constexpr uint32_t WG = 64;
    const OctCursor cursor{reinterpret_cast<uint2*>(dynamic_lds()) + threadIdx.x, WG};
    const OctDir dr = oct_dir(normalize(sr.d));
    const double wsd = (double)(1u << oc.world_log2);
    OctWalk w{{0.0, 0.0, 0.0}, 0, 0, 0, 0, 0};
    uint32_t solid_node = ~0u;
    int solid_code = 0;
    int status = OCT_IDLE;
    if (fresh) {
        status = OCT_MISS;
        if (oc.n_nodes != 0 && aabb_hit(D3{0.0, 0.0, 0.0}, D3{wsd, wsd, wsd}, sr, inv, t_min, closest)) {
            oct_cursor_start(nodes, oc, cursor);
            status = oct_walk_begin(nodes, oc, cursor, sr.o, dr, w, solid_node, solid_code);
        }
    } else if (resuming) {
        w = oct_walk_load();
        status = OCT_WALKING;
    }
    const uint32_t ws = 1u << oc.world_log2;
    for (uint32_t rounds = 0;; ++rounds) {
        const unsigned long long walking = __ballot(status == OCT_WALKING);
        if (walking == 0) break;
        // put the walk down when few lanes are left in it and the others have something to do (a finished query to
        // shade, or -- the caller's knowledge, in `may_suspend` -- a new path to start); never before every walk has made
        // OCT_SUSPEND_MIN_ROUNDS steps, so that a walk advances in every turn it is part of
        if (may_suspend && rounds >= (uint32_t)OCT_SUSPEND_MIN_ROUNDS && __popcll(walking) < OCT_SUSPEND_BELOW) break;
#ifdef MRT_OCT_DIAG
        if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1)
            atomicAdd(&g_oct_diag[0], (unsigned long long)__popcll(walking)), atomicAdd(&g_oct_diag[1], 1ull);
#endif
        if (status == OCT_WALKING) status = oct_step_solid(nodes, cursor, ws, dr, w, solid_node, solid_code);
    }
    suspended = status == OCT_WALKING;
    if (suspended) oct_walk_save(w);
    if (status == OCT_SOLID) {
        const D3 dp = w.cur - sr.o;
        const double t = sqrt(dot(dp, dp) / dot(sr.d, sr.d));
        if (t > t_min && t < closest) {  // strict, unlike the other shapes (fast_oct_tree/hittable.rs:28,48)
            closest = t, best_obj = r->object;
            best_sub = oct_node_at(nodes, solid_node).link * OCT_SUB + solid_code;
            best_pos = w.cur;
        }
    }
}

// Object::hit (hittable.rs:204-237) without the material: returns t and the sub-shape id.
// `inv_world` = 1/direction of the world ray; objects with the identity transform use the world
// ray as it is (the reference's M*(o+d) - M*o differs from d only by rounding).
// kind/p describe the shape; for a ConstantMedium (constant_medium.rs:35-93) bkind/p describe its
// boundary, `neg_inv_density` = -1/density, and one uniform is drawn at dimension `medium_dim`.
// The boundary queries of a medium and the plain query of a solid share ONE copy of the shape
// code (a loop of one or two rounds), which keeps the kernel's instruction footprint small.
template <int FEAT, class PP, class AP>
__device__ __forceinline__ bool leaf_hit(SceneRef sv, int kind, int bkind, int octree, PP p,
                                         double neg_inv_density, bool identity, AP to_object, const RayD& ray,
                                         D3 inv_world, double t_min, double t_max, const Rng& g, uint32_t medium_dim,
                                         uint32_t& oct_entry, double& t, int& sub, D3& pos, bool translation = false) {
#pragma clang fp contract(off)
    RayD sr = ray;
    D3 inv = inv_world;
    if (translation) {
        // a pure translation leaves the direction alone (the reference's M*(o+d) - M*o differs from d only by rounding)
        sr.o = {ray.o.x + to_object->r[0][3], ray.o.y + to_object->r[1][3], ray.o.z + to_object->r[2][3]};
    } else if (!identity) {
        sr = xform_ray(to_object, false, ray);
        inv = {rcp_fast(sr.d.x), rcp_fast(sr.d.y), rcp_fast(sr.d.z)};
    }
    if constexpr ((FEAT & FEAT_OCTREE) != 0) {
        if (kind == MRT_SHAPE_OCTREE)
            return octree_hit<(FEAT & FEAT_VOLUME) != 0>(sv.oct_nodes, sv.oct_voxels, sv.octrees[octree], sr, inv, t_min, t_max, g,
                                                         oct_entry, t, sub, pos);
    }
    bool medium = false;
    if constexpr ((FEAT & FEAT_MEDIUM) != 0) medium = kind == MRT_SHAPE_CONSTANT_MEDIUM;
    const int k = medium ? bkind : kind;
    double lo = medium ? -10000000000.0 : t_min;
    const double hi = medium ? 10000000000.0 : t_max;
    double t1 = 0.0, t2 = 0.0;
    int s1 = 0;
    int rounds = medium ? 2 : 1;
    if constexpr ((FEAT & FEAT_MEDIUM) != 0) {
        if (medium && bkind == MRT_SHAPE_RENDER_BOX) {
            if (!box_hit_twice(p, sr, inv, t1, t2)) return false;
            rounds = 0;
        }
    }
#pragma nounroll
    for (int c = 0; c < rounds; ++c) {
        double th;
        int sh;
        if (!solid_hit_kp<FEAT>(k, p, sr, inv, lo, hi, th, sh)) return false;
        if (c == 0) t1 = th, s1 = sh, lo = th + 0.0001;
        else t2 = th;
    }
    if (!medium) {
        t = t1, sub = s1;
        return true;
    }
    if constexpr ((FEAT & FEAT_MEDIUM) != 0) {
        sub = 0;
        if (t1 < t_min) t1 = t_min;
        if (t2 > t_max) t2 = t_max;
        if (t1 >= t2) return false;
        if (t1 < 0.0) t1 = 0.0;
        double len2 = fma(sr.d.z, sr.d.z, fma(sr.d.y, sr.d.y, sr.d.x * sr.d.x));
        double ray_length = sqrt(len2);
        double distance_inside = (t2 - t1) * ray_length;
        double hit_distance = neg_inv_density * log_unit(rng_f64(g, medium_dim));  // rand_scalar(0, 1) is the uniform itself
        if (hit_distance > distance_inside) return false;
        t = fma(hit_distance, rcp_fast(ray_length), t1);
        return true;
    }
    return false;
}
// Leaf test from an object record (the queries against the light list). A translated object takes the general
// transform here -- not the walk's shortcut: with the shortcut's code in the light query the headline kernel
// measured 2.4 % slower, and no shipped scene has a transformed light; surface_at is told (`shortcut` false).
template <int FEAT, class OP>
__device__ __forceinline__ bool object_hit(SceneRef sv, OP ob, const RayD& ray, D3 inv_world, double t_min,
                                           double t_max, const Rng& g, uint32_t medium_dim, uint32_t& oct_entry, double& t,
                                           int& sub, D3& pos) {
    return leaf_hit<FEAT>(sv, ob->kind, ob->gkind, ob->octree, &ob->p[0], ob->p[9], ob->identity == 1, &ob->to_object, ray,
                          inv_world, t_min, t_max, g, medium_dim, oct_entry, t, sub, pos);
}
// Leaf test of the wave-synchronous walk from a self-contained node record (scalar operands). `kind` and `flags`
// are the record's, which the walk holds already.
template <int FEAT, class RP>
__device__ __forceinline__ bool node_hit(SceneRef sv, RP r, int kind, int flags, const RayD& ray, D3 inv_world, double t_min,
                                         double t_max, const Rng& g, uint32_t medium_dim, uint32_t& oct_entry, double& t,
                                         int& sub, D3& pos) {
    const bool identity = (flags & NODE_IDENTITY) != 0;
    return leaf_hit<FEAT>(sv, kind, (flags >> NODE_KIND2_SHIFT) & 0xFF, r->octree, &r->p[0], r->p[9], identity, &r->to_object, ray,
                          inv_world, t_min, t_max, g, medium_dim, oct_entry, t, sub, pos, !identity && (flags & NODE_TRANSLATION) != 0);
}

// A ConstantMedium leaf of the wave-synchronous walk (constant_medium.rs:35-93), evaluated once or -- a span-1
// node pair folded into one node, NODE_TWICE -- twice: the reference then tests the leaf's box against the bound
// the first evaluation left and runs Object::hit again, with a fresh draw (bvh.rs:152-161,224-232). The
// object-space ray and the two boundary hits depend neither on the query range nor on the draw, so they are
// worked out once and the second evaluation only clamps and draws again (Cornell smoke: one of its two media is
// such a node; transform and boundary are 210 of the 400 instructions of an evaluation).
// `pass`: the lanes that passed the box test. Every lane of the wave may call this.
template <int FEAT, class RP>
__device__ __forceinline__ void medium_leaf_wave(RP r, int obj, int flags, bool pass, const RayD& ray, D3 inv_world, const Rng& g,
                                                 uint32_t& medium_ordinal, double& closest, int& best_obj, int& best_sub) {
#pragma clang fp contract(off)
    const double t_min = (double)0.001f;
    bool ok = pass;
    double b1 = 0.0, b2 = 0.0, ray_length = 0.0;
    {
        // everything the visit reads of the record -- transform, boundary parameters, density -- requested together
        // (the node's box is done with); read where it is used it is three round trips, one behind the other
        auto p = &r->p[0];
        asm volatile("" ::"s"(r->to_object.r[0][0]), "s"(r->to_object.r[0][1]), "s"(r->to_object.r[0][2]), "s"(r->to_object.r[0][3]),
                     "s"(r->to_object.r[1][0]), "s"(r->to_object.r[1][1]), "s"(r->to_object.r[1][2]), "s"(r->to_object.r[1][3]),
                     "s"(r->to_object.r[2][0]), "s"(r->to_object.r[2][1]), "s"(r->to_object.r[2][2]), "s"(r->to_object.r[2][3]),
                     "s"(p[0]), "s"(p[1]), "s"(p[2]), "s"(p[3]), "s"(p[4]), "s"(p[5]), "s"(p[9]));
    }
    if (pass) {
        RayD sr = ray;
        D3 inv = inv_world;
        const bool identity = (flags & NODE_IDENTITY) != 0;
        if (!identity && (flags & NODE_TRANSLATION) != 0) {
            sr.o = {ray.o.x + r->to_object.r[0][3], ray.o.y + r->to_object.r[1][3], ray.o.z + r->to_object.r[2][3]};
        } else if (!identity) {
            sr = xform_ray(&r->to_object, false, ray);
            inv = {rcp_fast(sr.d.x), rcp_fast(sr.d.y), rcp_fast(sr.d.z)};
        }
        const int k = (flags >> NODE_KIND2_SHIFT) & 0xFF;
        auto p = &r->p[0];
        if (k == MRT_SHAPE_RENDER_BOX) {
            ok = box_hit_twice(p, sr, inv, b1, b2);
        } else if ((FEAT & FEAT_SPHERE) != 0 && k == MRT_SHAPE_SPHERE) {
            ok = sphere_hit_twice(D3{p[1], p[2], p[3]}, p[0], sr, b1, b2);
        } else {  // nearest boundary hit in [-1e10, 1e10], then the nearest in [t1 + 1e-4, 1e10] (constant_medium.rs:36-44)
            double lo = -10000000000.0;
#pragma nounroll
            for (int c = 0; c < 2 && ok; ++c) {
                double th = 0.0;
                int sh;
                ok = solid_hit_kp<FEAT>(k, p, sr, inv, lo, 10000000000.0, th, sh);
                if (c == 0) b1 = th, lo = th + 0.0001;
                else b2 = th;
            }
        }
        ray_length = sqrt(fma(sr.d.z, sr.d.z, fma(sr.d.y, sr.d.y, sr.d.x * sr.d.x)));
    }
    const double neg_inv_density = r->p[9];
    const int evaluations = (flags & NODE_TWICE) != 0 ? 2 : 1;
#pragma nounroll
    for (int e = 0; e < evaluations; ++e) {
        if (e == 1)
            pass = pass && aabb_hit(D3{r->bmin[0], r->bmin[1], r->bmin[2]}, D3{r->bmax[0], r->bmax[1], r->bmax[2]}, ray, inv_world, t_min, closest);
        if (pass) {
            if (ok) {
                double t1 = b1, t2 = b2;
                if (t1 < t_min) t1 = t_min;
                if (t2 > closest) t2 = closest;
                if (!(t1 >= t2)) {
                    if (t1 < 0.0) t1 = 0.0;
                    const double distance_inside = (t2 - t1) * ray_length;
                    // rand_scalar(0, 1) is the uniform itself
                    const double hit_distance = neg_inv_density * log_unit(rng_f64(g, MRT_DIM_MEDIUM + medium_ordinal));
                    if (!(hit_distance > distance_inside)) {
                        closest = fma(hit_distance, rcp_fast(ray_length), t1);
                        best_obj = obj, best_sub = 0;
                    }
                }
            }
            medium_ordinal++;
        }
    }
}

// Leaf box of a MovingSphere-based object for the time window (t0, t1) the reference passes
// (bvh.rs:142-145,180; sphere.rs:175-186; hittable.rs:239-282).
template <class OP, class PP>
__device__ void dynamic_leaf_box(OP ob, PP p, bool identity, double t0, double t1, D3& mn, D3& mx) {
#pragma clang fp contract(off)
    D3 c0 = moving_center(p, t0), c1 = moving_center(p, t1);
    D3 r{p[0], p[0], p[0]};
    D3 lo0{c0.x - r.x, c0.y - r.y, c0.z - r.z}, hi0{c0.x + r.x, c0.y + r.y, c0.z + r.z};
    D3 lo1{c1.x - r.x, c1.y - r.y, c1.z - r.z}, hi1{c1.x + r.x, c1.y + r.y, c1.z + r.z};
    D3 smn{lo0.x < lo1.x ? lo0.x : lo1.x, lo0.y < lo1.y ? lo0.y : lo1.y, lo0.z < lo1.z ? lo0.z : lo1.z};
    D3 smx{hi0.x > hi1.x ? hi0.x : hi1.x, hi0.y > hi1.y ? hi0.y : hi1.y, hi0.z > hi1.z ? hi0.z : hi1.z};
    // (the identity matrix maps a finite corner to itself exactly)
    D3 a = identity ? smn : apply(&ob->to_world, smn), b = identity ? smx : apply(&ob->to_world, smx);
    mn = {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z};
    mx = {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z};
}

struct HitD {
    double t;
    int obj;
    int sub;   // RenderBox: face 0..5; octree: voxel * 6 + normal code
    D3 pos;    // octree only: the DDA's hit position (the record's position, fast_oct_tree/hittable.rs:29-36)
};

// World::nearest_hit (world.rs:112-114) -> BvhTreeNode::hit (bvh.rs:135-167), stackless over the
// pre-order node arrays. Visiting left before right with t_max = left.t and preferring the
// right result on ties is the same as keeping one running closest t and accepting t <= closest.
//
// A span-1 node pair folded into one node (NODE_TWICE) evaluates its object a second time after
// re-testing the box against the updated bound, with a fresh medium draw (bvh.rs:224-232).
//
// Per-lane variant: every lane walks its own node sequence (large trees, incoherent rays).
struct LaneWalk {
    double closest;
    int best_obj, best_sub;
    D3 best_pos;
    uint32_t medium_ordinal, oct_entry;
};
// One node visit of the walk: the node's box test against the running bound and, for a leaf that
// passes, the object's intersection test(s). Returns whether the box test passed.
// The leaf's shape record `lr` comes with the node (same index, requested together with it), so a visit
// costs one round trip to memory; the object record itself is only touched for objects with a transform.
template <int FEAT>
__device__ __forceinline__ bool lane_visit(SceneRef sv, const DNode& nd, const DLeafRec& lr, const RayD& ray, D3 inv_d,
                                           const Rng& g, LaneWalk& w, double box_bound) {
    const double t_min = (double)0.001f;
    const int obj = nd.object;
    const int flags = nd.flags;
    D3 mn{nd.bmin[0], nd.bmin[1], nd.bmin[2]};
    D3 mx{nd.bmax[0], nd.bmax[1], nd.bmax[2]};
    const DObjRec* ob = sv.objects + (obj >= 0 ? obj : 0);
    const int4 head{lr.kind, lr.gkind, lr.identity, lr.octree};
    const double* p = lr.p;
    const bool identity = head.z == 1, translation = head.z == 2;
    if constexpr ((FEAT & FEAT_MOVING) != 0) {
        if (flags & NODE_DYNAMIC_BOX) dynamic_leaf_box(ob, p, identity, t_min, box_bound, mn, mx);
    }
    // (`box_bound`: the running bound, except for a leaf visit_candidates tests at its cap)
    if (!aabb_hit(mn, mx, ray, inv_d, t_min, box_bound)) return false;
    if (obj >= 0) {
        bool is_medium = false;
        if constexpr ((FEAT & FEAT_MEDIUM) != 0) is_medium = head.x == MRT_SHAPE_CONSTANT_MEDIUM;
        // (only stochastic shapes -- media, octrees with volume voxels -- are ever marked)
        const int rounds = ((FEAT & (FEAT_MEDIUM | FEAT_VOLUME)) != 0 && (flags & NODE_TWICE)) ? 2 : 1;
#pragma nounroll
        for (int round = 0; round < rounds; ++round) {
            if (round == 1 && !aabb_hit(mn, mx, ray, inv_d, t_min, w.closest)) break;
            double t;
            int sub;
            if (leaf_hit<FEAT>(sv, head.x, head.y, head.w, p, p[9], identity, &ob->to_object, ray, inv_d, t_min, w.closest, g,
                               MRT_DIM_MEDIUM + w.medium_ordinal, w.oct_entry, t, sub, w.best_pos, translation)) {
                w.closest = t, w.best_obj = obj, w.best_sub = sub;
            }
            if (is_medium) w.medium_ordinal++;
        }
    }
    return true;
}
// Per-ray constants of the candidate search's slab tests, t = bound * inv - origin * inv, in single precision.
// The search only has to mark a SUPERSET of the leaves the ray's line meets (every marked leaf is tested again
// with the walk's own arithmetic), so its slab tests run in f32 -- half the lane reads, no conversions of the f32
// boxes, two slabs per instruction -- and every slab interval is widened by a bound on what f32 rounds away. With
// u = 2^-24, B the largest |bound| of the node's children on the axis and V their largest |velocity bound|: a
// child bound at the ray's time (tau rounded, the product rounded, the sum rounded) is within u (B + 3 |tau| V) of
// its exact value, inv and origin * inv within u of theirs, and the fused t rounds once more, so
// |t32 - t| <= 3 u S (1 + small) with S = (B + 3 |tau| V) |inv| + |origin * inv|; widening by CAND_SLACK * S leaves
// 1.5 u S for the rounding of S itself and of the widened ends. (The slack matters: a bounced ray starts ON a
// sphere, whose slab interval ends at t = 0 just below the query's 0.001; one slack for the whole tree -- its
// largest bound is the ground sphere's 2000 -- marked that sphere for most rays, 12 % of the kernel.)
#ifndef MRT_CAND_SLACK_U
#define MRT_CAND_SLACK_U 4.5f  // (a build with 0 must fail tests/test_candidate_search.py)
#endif
constexpr float CAND_SLACK = MRT_CAND_SLACK_U * 0x1p-24f;
struct CandRay {
    float inv[3], oi[3];
    float tau;  // the ray's time
    float cap;  // leaves whose boxes the line enters beyond this t are not wanted (FLT_MAX: all of them)
};
typedef float v4f __attribute__((ext_vector_type(4)));
// The constants of a ray for the search; `ok` false: a component is outside the range where the f32 bound holds
// (the ray takes the plain walk).
__device__ __forceinline__ CandRay cand_ray(const RayD& ray, D3 inv_d, float cap, bool& ok) {
    const D3 oi{ray.o.x * inv_d.x, ray.o.y * inv_d.y, ray.o.z * inv_d.z};
    const double imin = fmin(fabs(inv_d.x), fmin(fabs(inv_d.y), fabs(inv_d.z)));
    const double imax = fmax(fabs(inv_d.x), fmax(fabs(inv_d.y), fabs(inv_d.z)));
    const double omax = fmax(fabs(oi.x), fmax(fabs(oi.y), fabs(oi.z)));
    // f32's normal range with room for the products; CAND_MIN_BOUND * CAND_SLACK * 1e-18 is still far above the
    // smallest f32 step, which bounds what a result in the denormal range loses
    ok = imin > 1e-18 && imax < 1e18 && omax < 1e18 && fabs(ray.time) < 1e9;
    return CandRay{{(float)inv_d.x, (float)inv_d.y, (float)inv_d.z}, {(float)oi.x, (float)oi.y, (float)oi.z}, (float)ray.time,
                   cap < FLT_MAX ? cap : FLT_MAX};
}
// LDS state of the candidate search. Per lane (columns of [row][threads per workgroup] arrays): a bit per
// tree node, all zero between queries, in rows 0 .. CAND_WORDS - 1, and in row CAND_WORDS one bit per row
// that holds marks. Per wave: the work queue of the spatial search, CAND_QUEUE entries.
struct CandLds {
    LDS_AS uint32_t* bits;   // this lane's column
    LDS_AS uint16_t* queue;  // this wave's queue
    uint32_t stride;         // threads per workgroup
};
constexpr int CAND_QUEUE = CAND_STACK * 64;
// Phase 1 of the candidate search for all lanes of a wave at once (`asking` lanes bring the ray constants
// `cr`): walks the spatial tree and marks, in the asking lane's bit column, every leaf whose box the ray's line
// meets. A lane walking alone visits 6.4 nodes of the random-spheres tree on average, but the wave waits for
// its longest walk, 14.5 rounds. Which leaves get marked does not depend on the order of the visits, so the
// visits of all lanes go through one queue of (lane, node) entries per wave and every round the 64 lanes take
// the 64 newest entries, whoever's they are: a lane fetches the constants of the ray it works for from that
// ray's lane (ds_bpermute), tests the node's four children, ORs leaf marks into that lane's column and the
// wave appends the inner children that were hit. Returns false when the queue could overflow (the marks are
// incomplete then; the caller clears them and takes the plain walk). Every lane of the wave must call this.
__device__ bool wave_candidate_search(SceneRef sv, const CandRay& cr, bool asking, CandLds lds, Stamps& st) {
    const uint32_t lane = __lane_id();
    LDS_AS uint16_t* queue = lds.queue;
    LDS_AS uint32_t* column0 = lds.bits - lane;  // column of lane 0 of this wave
    const unsigned long long ask = __ballot(asking);
    uint32_t n_items = (uint32_t)__popcll(ask);  // wave-uniform
    if (asking)  // every asking lane starts at the root (node 0)
        queue[__builtin_amdgcn_mbcnt_hi((uint32_t)(ask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ask, 0u))] = (uint16_t)(lane << 10);
    bool complete = true;
    while (n_items > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t take = n_items < 64u ? n_items : 64u;
        if (n_items + 3u * take > (uint32_t)sv.cand_queue) {
            complete = false;
            break;
        }
        const bool work = lane < take;
        uint32_t item = 0;
        if (work) item = queue[n_items - 1u - lane];
        n_items -= take;
        const uint32_t owner = item >> 10, node = item & 1023u;
        // One node = one batch of independent loads (the whole first line, then the velocity line of a
        // node with moving children), requested before anything is tested: a load issued inside a
        // per-child branch would add its own round trip to the chain of node fetches.
        const DCand4* nd = sv.cand_nodes + node;
        const int4 ch4 = *reinterpret_cast<const int4*>(nd->child);
        v4f lo4[3], hi4[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo4[a] = *reinterpret_cast<const v4f*>(nd->lo[a]);
            hi4[a] = *reinterpret_cast<const v4f*>(nd->hi[a]);
        }
        // (the velocity line too, whether or not the node has a moving child: asked for only after the flag had
        // arrived, it was a second round trip in a loop that is bound by the chain of its waits)
        v4f vlo4[3], vhi4[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            vlo4[a] = *reinterpret_cast<const v4f*>(nd->vlo[a]);
            vhi4[a] = *reinterpret_cast<const v4f*>(nd->vhi[a]);
        }
        // the constants of the ray this lane works for
        const int from = (int)owner;
        const float inv[3] = {__shfl(cr.inv[0], from, 64), __shfl(cr.inv[1], from, 64), __shfl(cr.inv[2], from, 64)};
        const float oi[3] = {__shfl(cr.oi[0], from, 64), __shfl(cr.oi[1], from, 64), __shfl(cr.oi[2], from, 64)};
        const float tau = __shfl(cr.tau, from, 64);
        const float cap = __shfl(cr.cap, from, 64);
        if (work) STAMP_COUNT(st, 0);
        STAMP_COUNT_WAVE(st, 4);
        // a child's box at the ray's time: lo + min(tau * vlo, tau * vhi) .. hi + max(tau * vlo, tau * vhi) (a static
        // child has zero velocities and keeps its box exactly)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const v4f pl = (v4f)(tau) * vlo4[a], ph = (v4f)(tau) * vhi4[a];
            lo4[a] += __builtin_elementwise_min(pl, ph);
            hi4[a] += __builtin_elementwise_max(pl, ph);
        }
        // whether the ray's line over [0.001, FLT_MAX] can meet the boxes of the four children at the ray's time (all
        // three slabs jointly); an absent child has an inside-out box (lo = inf, hi = -inf) and never passes
        v4f lo = (v4f)(0.001f), hi = (v4f)(cap);
        const float4 scale = *reinterpret_cast<const float4*>(nd->bound);  // bound[3], vel
        const float tv = 3.f * __builtin_fabsf(tau) * scale.w;
        const float bnd[3] = {scale.x + tv, scale.y + tv, scale.z + tv};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const v4f t0 = __builtin_elementwise_fma(lo4[a], (v4f)(inv[a]), (v4f)(-oi[a]));
            const v4f t1 = __builtin_elementwise_fma(hi4[a], (v4f)(inv[a]), (v4f)(-oi[a]));
            const float slack = CAND_SLACK * __builtin_fmaf(bnd[a], __builtin_fabsf(inv[a]), __builtin_fabsf(oi[a]));
            lo = __builtin_elementwise_max(lo, __builtin_elementwise_min(t0, t1) - (v4f)(slack));
            hi = __builtin_elementwise_min(hi, __builtin_elementwise_max(t0, t1) + (v4f)(slack));
        }
        const int child[4] = {ch4.x, ch4.y, ch4.z, ch4.w};
        LDS_AS uint32_t* column = column0 + owner;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool meets = work && lo[k] <= hi[k] && child[k] != -1;
            if (meets && child[k] < 0) {
                // a leaf: mark it in the bit column of the ray's lane (LDS ORs that return nothing)
                const int leaf = -2 - child[k];
                (void)__hip_atomic_fetch_or(column + (leaf >> 5) * lds.stride, 1u << (leaf & 31), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_or(column + CAND_WORDS * lds.stride, 1u << (leaf >> 5), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // an inner node: the wave appends it to the queue
            const bool push = meets && child[k] >= 0;
            const unsigned long long pm = __ballot(push);
            if (push)
                queue[n_items + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u))] =
                    (uint16_t)((owner << 10) | (uint32_t)child[k]);
            n_items += (uint32_t)__popcll(pm);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return complete;
}
// Candidate search for large trees (host side: build_candidate_tree). Phase 1 (wave_candidate_search) has
// marked every leaf whose tight box the ray's line meets; this is phase 2, the reference's walk over the
// marked leaves (ascending index = pre-order) with the walk's own node code. Gives the walk's result.
// With `discard` the marks are only cleared.
template <int FEAT>
__device__ bool visit_candidates(SceneRef sv, const RayD& ray, D3 inv_d, const Rng& g, HitD& out, CandLds lds, bool discard,
                                 Stamps& st, float cap = FLT_MAX, bool* retry = nullptr) {
    LDS_AS uint32_t* bits = lds.bits;
    static_assert(CAND_WORDS <= 32, "one summary bit per word");
    uint32_t touched = bits[CAND_WORDS * lds.stride];  // which words of the bit array hold marks
    bits[CAND_WORDS * lds.stride] = 0;
    if (discard) {
        while (touched) {
            bits[(__ffs((int)touched) - 1) * lds.stride] = 0;
            touched &= touched - 1;
        }
        return false;
    }
    STAMP_COUNT(st, 3);
    STAMP_COUNT_WAVE(st, 6);
    LaneWalk w{(double)FLT_MAX, -1, 0, {0.0, 0.0, 0.0}, 0, 0};
    // The marked nodes are known before any of them is visited, so the next node's record is
    // requested while the current one is processed (the walk is bound by load latency).
    int wd = 0;
    uint32_t m = 0;
    auto next_marked = [&]() -> int {  // ascending node indices; clears the words it reads
        if (m == 0) {
            if (touched == 0) return -1;
            wd = __ffs((int)touched) - 1;
            touched &= touched - 1;
            m = bits[wd * lds.stride];
            bits[wd * lds.stride] = 0;
        }
        const int idx = wd * 32 + __ffs((int)m) - 1;
        m &= m - 1;
        return idx;
    };
    // The walk reaches a leaf iff the boxes of all its ancestors passed when the walk came to them. Those
    // boxes are nested (an inner box is the union of its children's, bvh.rs:280-286) and the running bound
    // only shrinks, and Aabb::hit is monotone in both: if the innermost ancestor -- the leaf's parent -- passes
    // against the bound it met, every outer one passed against its own, larger bound. A leaf with a static
    // box lies inside its parent's box and is tested later, so its own test implies the parent's too. That
    // leaves one extra test: the parent of a leaf whose box depends on the query (MovingSphere, built into
    // the parents at the shutter times only), once per parent, against the bound at the time the walk enters
    // the parent -- which is the bound before the parent's first leaf, because the marked leaves come in
    // pre-order.
    int last_parent = -1;
    bool last_parent_pass = true;
#ifdef MRT_STAMPS
    int diag_full = 0;
#endif
    int idx = next_marked();
    DNode nd{};
    DLeafRec lr{};
    if (idx >= 0) nd = sv.nodes[idx], lr = sv.lane_leaf[idx];
    while (idx >= 0) {
        const int idx_next = next_marked();
        DNode nd_next{};
        DLeafRec lr_next{};
        if (idx_next >= 0) nd_next = sv.nodes[idx_next], lr_next = sv.lane_leaf[idx_next];
        STAMP_COUNT(st, 2);
        STAMP_COUNT_WAVE(st, 5);
#ifdef MRT_STAMPS
        if (lr.kind == MRT_SHAPE_SPHERE && lr.identity == 1) STAMP_COUNT(st, 11); else diag_full++;
#endif
        bool reach = true, above = false;  // `above`: a query-dependent leaf met with the bound above the search's cap
        double box_bound = w.closest;
        if constexpr ((FEAT & FEAT_MOVING) != 0) {
            if (nd.flags & NODE_DYNAMIC_BOX) {
                // The search ran under `cap`, an upper bound on this query's answer, and left out the leaves whose
                // boxes the line enters beyond it. Their hits, all beyond the cap, cannot be the answer, but the walk
                // would have visited them and they would have moved its running bound, on which the box of THIS
                // leaf (and the parent test in front of it) depends. Once the bound is at or below the cap they are
                // out of the picture -- a hit beyond the bound is never accepted -- and the bound here is the walk's.
                // Above the cap the walk's bound is somewhere in [cap, bound here] (it can only be lower: it saw more
                // leaves). For an untransformed MovingSphere both tests are monotone in the bound -- its box for the
                // window (t_min, bound) only grows with it, and so does the range of Aabb::hit -- so they are made AT
                // THE CAP: passing there, they pass for the walk, and the bound kept here stays an upper bound of the
                // walk's (a hit accepted here beyond the cap is replaced before the query ends). Failing there, or
                // with a transformed sphere (two mapped corners, hittable.rs:239-282: not monotone), nothing is known
                // and the query is put to the uncapped search (see nearest_hit_lane).
                above = w.closest > (double)cap;
                box_bound = above ? (double)cap : w.closest;
#ifdef MRT_MUTANT_NO_RETRY  // (a build with this must fail the renders of tests/test_candidate_search.py)
                above = false, box_bound = w.closest;
#endif
                if (above && lr.identity != 1) *retry = true;
                if (nd.parent != last_parent) {
                    last_parent = nd.parent;
                    last_parent_pass = true;
                    if (nd.parent >= 0) {
                        STAMP_COUNT(st, 1);
                        const DNode pn = sv.nodes[nd.parent];
                        last_parent_pass = aabb_hit(D3{pn.bmin[0], pn.bmin[1], pn.bmin[2]}, D3{pn.bmax[0], pn.bmax[1], pn.bmax[2]}, ray,
                                                    inv_d, (double)0.001f, box_bound);
                        if (above && !last_parent_pass) *retry = true;
                    }
                }
                reach = last_parent_pass;
            }
        }
        if (reach) {
            const bool box_passed = lane_visit<FEAT>(sv, nd, lr, ray, inv_d, g, w, box_bound);
            if constexpr ((FEAT & FEAT_MOVING) != 0) {
                if (above && !box_passed) *retry = true;
            }
        }
        if constexpr ((FEAT & FEAT_MOVING) != 0) {
            if (cap < FLT_MAX && *retry) {  // nothing of this pass is kept: clear the marks that are left
                if (idx_next >= 0) while (next_marked() >= 0) {}
                return false;
            }
        }
        idx = idx_next;
        nd = nd_next;
        lr = lr_next;
    }
#ifdef MRT_STAMPS
    for (int k = 1; k < 64; ++k) {  // rounds the wave would need if visits of untransformed static spheres cost nothing
        if (__ballot(diag_full >= k) == 0) break;
        STAMP_COUNT_WAVE(st, 10);
    }
#endif
    out.t = w.closest;
    out.obj = w.best_obj;
    out.sub = w.best_sub;
    out.pos = w.best_pos;
    return w.best_obj >= 0;
}
// Per-lane nearest hit for large trees: the candidate search where the scene has its spatial tree, else (and
// for the rays the search does not take) the plain walk. Every lane of the wave must call this; `live` marks
// the lanes with a ray.
template <int FEAT>
__device__ bool nearest_hit_lane(SceneRef sv, const RayD& ray, const Rng& g, bool live, HitD& out, Stamps& st,
                                 CandLds lds = CandLds{nullptr, nullptr, 0}, float cap = INFINITY, bool* retry = nullptr) {
    // (`stride`, not a pointer, says whether the kernel has the search's LDS arrays: LDS address 0 is thread 0's
    // column, and the test must come out the same for the whole wave)
    if (sv.n_cand_nodes > 0 && lds.stride != 0) {
        // (a zero or non-finite direction component makes slab bounds undefined: those rays take the walk)
        const double dmin = fmin(fabs(ray.d.x), fmin(fabs(ray.d.y), fabs(ray.d.z)));
        const double dmax = fmax(fabs(ray.d.x), fmax(fabs(ray.d.y), fabs(ray.d.z)));
        // (so do origins so far out that origin * inv loses the digits the boxes' safety margin covers)
        const double omax = fmax(fabs(ray.o.x), fmax(fabs(ray.o.y), fabs(ray.o.z)));
        D3 inv_d{rcp_fast(ray.d.x), rcp_fast(ray.d.y), rcp_fast(ray.d.z)};
        bool in_range;
        // `cap`: an upper bound on this query's answer that the caller knows (INFINITY if none): the search leaves out
        // what lies beyond it. With leaves whose boxes depend on the running bound that is not always enough to
        // reproduce the walk (visit_candidates); `*retry` then comes back true, the lane has no answer, and the caller
        // asks again without a cap -- in the kernel loop's next turn, where the search runs for the whole wave anyway.
        const CandRay cr = cand_ray(ray, inv_d, retry ? cap : INFINITY, in_range);
        const bool search = live && dmin > 0.0 && dmax < INFINITY && omax < 1e6 && in_range;
        REGION_BEGIN(search_t);
        const bool complete = wave_candidate_search(sv, cr, search, lds, st);
        REGION_END(st, ST_CAND_SEARCH, search_t);
        if (search) {
            bool again = false;
            const bool hit = visit_candidates<FEAT>(sv, ray, inv_d, g, out, lds, !complete, st, cr.cap, &again);
            if (complete) {
                if (again) {
                    *retry = true;
                    STAMP_COUNT(st, 8);
                    return false;
                }
                return hit;
            }
        }
    }
    if (!live) return false;
    LaneWalk w{(double)FLT_MAX, -1, 0, {0.0, 0.0, 0.0}, 0, 0};
    D3 inv_d{rcp_fast(ray.d.x), rcp_fast(ray.d.y), rcp_fast(ray.d.z)};
    int i = 0;
    const int n = sv.n_nodes;
    while (i < n) {
        const DNode nd = sv.nodes[i];
        const DLeafRec lr = sv.lane_leaf[i];
        i = lane_visit<FEAT>(sv, nd, lr, ray, inv_d, g, w, w.closest) ? i + 1 : nd.skip;
    }
    out.t = w.closest;
    out.obj = w.best_obj;
    out.sub = w.best_sub;
    out.pos = w.best_pos;
    return w.best_obj >= 0;
}

// Wave-synchronous variant for small trees: the whole wave visits the nodes in pre-order with
// ONE node index, so a node's record (box, links, the leaf's shape parameters and transform)
// is read with scalar loads (SGPR operands, no per-lane scene loads) and the shape-kind switch
// is a scalar branch. A lane that fails a box test sits out until the walk leaves that subtree
// (`resume`); a subtree no lane wants is skipped for the whole wave by a ballot. Must be called
// by every lane of the wave; lanes with live == false only follow along.
// `walking` (segment kernels of scenes whose first leaf is a solid-voxel octree, SceneView::oct_first): in, the lanes
// whose octree walk was suspended in the kernel loop's previous turn; out, the lanes whose walk is suspended now and that
// have no answer yet. `may_suspend`: wave-uniform, the caller has use for the lanes that are not walking.
template <int FEAT>
__device__ bool nearest_hit_wave(SceneRef sv, const RayD& ray, const Rng& g, bool live, HitD& out, Stamps& st, bool* walking = nullptr,
                                 bool may_suspend = false) {
    const double t_min = (double)0.001f;
    double closest = (double)FLT_MAX;
    int best_obj = -1, best_sub = 0;
    D3 best_pos{0.0, 0.0, 0.0};
    uint32_t medium_ordinal = 0, oct_entry = 0;
    D3 inv_d{rcp_fast(ray.d.x), rcp_fast(ray.d.y), rcp_fast(ray.d.z)};
    const int n = sv.n_wave_nodes;
    auto recs = as_const(sv.node_recs);
    int resume = live ? 0 : n;
    int i = 0;
    if constexpr ((FEAT & FEAT_OCTREE) != 0 && (FEAT & FEAT_VOLUME) == 0) {
        if (walking != nullptr && sv.oct_first != 0) {
            // the first node: a leaf with a solid-voxel octree, whose walk can stop between steps
            auto r = recs;
            const int flags = r->flags;
            const bool resuming = live && *walking;
            const bool fresh = live && !resuming &&
                               aabb_hit(D3{r->bmin[0], r->bmin[1], r->bmin[2]}, D3{r->bmax[0], r->bmax[1], r->bmax[2]}, ray, inv_d, t_min, closest);
            bool suspended = false;
            REGION_BEGIN(leaf0);
            octree_first_leaf(sv, r, flags, ray, inv_d, fresh, resuming, may_suspend, t_min, closest, best_obj, best_sub, best_pos, suspended);
            REGION_END(st, ST_LEAF, leaf0);
            *walking = suspended;
            if (suspended) resume = n;  // no answer in this turn: the lane sits out the rest of the walk
            i = 1;
        }
    }
    const bool axis_parallel = ray.d.x == 0.0 || ray.d.y == 0.0 || ray.d.z == 0.0;
    // (wave-uniform) the node at hand is a NODE_TWICE node on its second evaluation. The second evaluation is one
    // more turn of this loop on the same node -- box test against the updated bound by the lanes that passed the
    // first, then the leaf -- so that the loop body holds one box test and one leaf test.
    bool second = false;
    while (i < n) {
        auto r = recs + i;  // wave-uniform: every field below is a scalar load
        const int obj = r->object;
        const int flags = r->flags;
        const int skip = r->skip;
        const int kind_at_top = r->kind;
        const double a0_at_top = r->p[0], a1_at_top = r->p[1], b0_at_top = r->p[2], b1_at_top = r->p[3], k_at_top = r->p[4];
        asm volatile("" ::"s"(obj), "s"(kind_at_top), "s"(a0_at_top), "s"(a1_at_top), "s"(b0_at_top), "s"(b1_at_top), "s"(k_at_top));
        if ((flags & NODE_BOX_IMPLIED) != 0) {
            // An untransformed rectangle or sphere: its shape test decides by itself, and all it reads is its
            // five (four) parameters. Left to the compiler the leaf is three round trips to the scalar cache, one behind the
            // other (links and flags; kind and transform class; parameters) around two dozen instructions.
            if (i >= resume) {
                const int kind = kind_at_top;
                const double a0 = a0_at_top, a1 = a1_at_top, b0 = b0_at_top, b1 = b1_at_top, k = k_at_top;
                double t;
                bool hit;
                if ((FEAT & FEAT_SPHERE_LEAF) != 0 && kind == MRT_SHAPE_SPHERE)  // (parameters: radius, centre)
                    hit = sphere_hit(D3{a1, b0, b1}, a0, ray, t_min, closest, t);
                else
                    hit = kind == MRT_SHAPE_XY_RECT   ? rect_hit<2>(a0, a1, b0, b1, k, ray, inv_d, t_min, closest, t)
                          : kind == MRT_SHAPE_XZ_RECT ? rect_hit<1>(a0, a1, b0, b1, k, ray, inv_d, t_min, closest, t)
                                                      : rect_hit<0>(a0, a1, b0, b1, k, ray, inv_d, t_min, closest, t);
                // (A ray lying in the rectangle's plane "hits" it at t = 0 * inf = NaN, which passes every
                // comparison of the shape test -- so does a sphere's NaN discriminant; the box test is what rejects
                // such a ray, so a NaN hit still takes it. So does a sphere's hit by a ray with a zero direction
                // component: such a ray can run along a face of the box, outside by one ulp, and still graze the
                // sphere within rounding. A rectangle's test is exact for it: the coordinate does not move.)
                if (hit && (t != t || ((FEAT & FEAT_SPHERE_LEAF) != 0 && kind == MRT_SHAPE_SPHERE && axis_parallel)))
                    hit = aabb_hit(D3{r->bmin[0], r->bmin[1], r->bmin[2]}, D3{r->bmax[0], r->bmax[1], r->bmax[2]}, ray, inv_d, t_min, closest);
                if (hit) closest = t, best_obj = obj, best_sub = 0;
            }
            i = i + 1;
            continue;
        }
        bool pass = false;
        if (i >= resume) {
            bool dynamic = false;
            if constexpr ((FEAT & FEAT_MOVING) != 0) dynamic = (flags & NODE_DYNAMIC_BOX) != 0;
            if (dynamic) {
                D3 mn, mx;
                dynamic_leaf_box(as_const(sv.objects) + obj, r->p, (flags & NODE_IDENTITY) != 0, t_min, closest, mn, mx);
                pass = aabb_hit(mn, mx, ray, inv_d, t_min, closest);
            } else {
                pass = aabb_hit(D3{r->bmin[0], r->bmin[1], r->bmin[2]}, D3{r->bmax[0], r->bmax[1], r->bmax[2]}, ray, inv_d, t_min, closest);
            }
            if (!pass) resume = skip;
        }
        if (!__any(pass)) {
            i = skip;
            second = false;
            continue;
        }
        if (obj >= 0) {
            REGION_BEGIN(leaf);
            bool is_medium = false;
            if constexpr ((FEAT & FEAT_MEDIUM) != 0) is_medium = kind_at_top == MRT_SHAPE_CONSTANT_MEDIUM;
            if (is_medium) {
                if constexpr ((FEAT & FEAT_MEDIUM) != 0)
                    medium_leaf_wave<FEAT>(r, obj, flags, pass, ray, inv_d, g, medium_ordinal, closest, best_obj, best_sub);
            } else if (pass) {
                double t;
                int sub;
                if (node_hit<FEAT & ~FEAT_MEDIUM>(sv, r, kind_at_top, flags, ray, inv_d, t_min, closest, g, 0, oct_entry, t, sub, best_pos)) {
                    closest = t, best_obj = obj, best_sub = sub;
                }
            }
            REGION_END(st, is_medium ? ST_MEDIUM_TAIL : ST_LEAF, leaf);
            // (of the shapes that are not media only octrees with volume voxels are stochastic and ever marked)
            if constexpr ((FEAT & FEAT_VOLUME) != 0) {
                if (!is_medium && (flags & NODE_TWICE) != 0 && !second) {
                    second = true;
                    continue;
                }
            }
        }
        second = false;
        i = i + 1;
    }
    out.t = closest;
    out.obj = best_obj;
    out.sub = best_sub;
    out.pos = best_pos;
    return best_obj >= 0;
}

// ============================================================== surface data at a hit
struct Surface {
    D3 pos_obj;   // hit.position as the shape computed it (object space)
    D3 n_obj;     // geometric normal in object space (never flipped, hittable.rs:385-391)
    D3 dir_obj;   // object-space ray direction
    D3 org_obj;   // object-space ray origin
    double t;
    double time;  // object-space ray time (= world ray time)
    int voxel;  // octree hits: index of the voxel record, else -1
    int voxel_code;  // octree hits: OCT_SUB code (own solid material / edge material / volume)
    int sub;
    bool front_face;  // as the material sees it (object space; `true` for media, constant_medium.rs:66-72)
};
__device__ __forceinline__ void sphere_uv(D3 p, double& u, double& v) {  // sphere.rs:111-115
    double theta = acos(-p.y);
    double phi = atan2(-p.z, p.x) + M_PI;
    // (the two quotients as the division gives them, by the constants' reciprocals; theta and phi are 0 .. 2 pi)
    u = div_by_reciprocal(phi, 2.0 * M_PI, 1.0 / (2.0 * M_PI));
    v = div_by_reciprocal(theta, M_PI, 1.0 / M_PI);
}
template <int AXIS>
__device__ __forceinline__ void rect_uv(double a0, double a1, double b0, double b1, const Surface& s, double& u,
                                        double& v) {
    constexpr int IA = AXIS == 0 ? 1 : 0;
    constexpr int IB = AXIS == 2 ? 1 : 2;
    double a = comp(s.org_obj, IA) + s.t * comp(s.dir_obj, IA);
    double b = comp(s.org_obj, IB) + s.t * comp(s.dir_obj, IB);
    u = (a - a0) / (a1 - a0);
    v = (b - b0) / (b1 - b0);
}
// Texture coordinates of the hit; evaluated only for textures that read them.
template <int FEAT, class OP>
__device__ void surface_uv(OP sh, const Surface& s, double& u, double& v) {
    auto p = sh->p;
    u = 0.0, v = 0.0;
    switch (sh->kind) {
        case MRT_SHAPE_SPHERE:
        case MRT_SHAPE_MOVING_SPHERE:
            if constexpr ((FEAT & FEAT_SPHERE) != 0) sphere_uv(s.n_obj, u, v);
            break;
        case MRT_SHAPE_XY_RECT: rect_uv<2>(p[0], p[1], p[2], p[3], s, u, v); break;
        case MRT_SHAPE_XZ_RECT: rect_uv<1>(p[0], p[1], p[2], p[3], s, u, v); break;
        case MRT_SHAPE_YZ_RECT: rect_uv<0>(p[0], p[1], p[2], p[3], s, u, v); break;
        case MRT_SHAPE_RENDER_BOX:
            if (s.sub < 2) rect_uv<2>(p[0], p[3], p[1], p[4], s, u, v);
            else if (s.sub < 4) rect_uv<1>(p[0], p[3], p[2], p[5], s, u, v);
            else rect_uv<0>(p[1], p[4], p[2], p[5], s, u, v);
            break;
        default: break;  // ConstantMedium: uv (0,0)
    }
}
template <int FEAT, class OP>
__device__ void surface_at(OP ob, const RayD& ray, double t, int sub, D3 oct_pos, Surface& s, bool shortcut = true) {
    // (the object-space ray the leaf test used: an untransformed object sees the world ray, a translated one the
    // world direction from the moved origin, see leaf_hit -- t and the hit point then come from ONE direction)
    RayD sr = ray;
    const int xf = ob->identity;
    if (xf == 2 && shortcut) sr.o = {ray.o.x + ob->to_object.r[0][3], ray.o.y + ob->to_object.r[1][3], ray.o.z + ob->to_object.r[2][3]};
    else if (xf != 1) sr = xform_ray(&ob->to_object, false, ray);
    OP sh = ob;
    auto p = sh->p;
    s.pos_obj = ray_at(sr, t);
    s.dir_obj = sr.d;
    s.org_obj = sr.o;
    s.t = t;
    s.time = sr.time;
    s.voxel = -1;
    s.voxel_code = 0;
    s.sub = sub;
    s.n_obj = {1.0, 0.0, 0.0};
    switch (sh->kind) {
        case MRT_SHAPE_SPHERE:
            if constexpr ((FEAT & FEAT_SPHERE) != 0) s.n_obj = (s.pos_obj - D3{p[1], p[2], p[3]}) * rcp_fast(p[0]);
            break;
        case MRT_SHAPE_MOVING_SPHERE:
            if constexpr ((FEAT & FEAT_MOVING) != 0) s.n_obj = (s.pos_obj - moving_center(p, sr.time)) * rcp_fast(p[0]);
            break;
        case MRT_SHAPE_XY_RECT: s.n_obj = {0.0, 0.0, p[5]}; break;
        case MRT_SHAPE_XZ_RECT: s.n_obj = {0.0, p[5], 0.0}; break;
        case MRT_SHAPE_YZ_RECT: s.n_obj = {p[5], 0.0, 0.0}; break;
        case MRT_SHAPE_RENDER_BOX: {  // render_box.rs:21-93: "+" faces unflipped, "-" faces flipped
            double flip = (sub & 1) ? -1.0 : 1.0;
            s.n_obj = {sub >= 4 ? flip : 0.0, (sub == 2 || sub == 3) ? flip : 0.0, sub < 2 ? flip : 0.0};
            break;
        }
        case MRT_SHAPE_OCTREE: {  // fast_oct_tree/hittable.rs:19-36: position from the walk, axis normal
            if constexpr ((FEAT & FEAT_OCTREE) == 0) break;
            s.voxel_code = sub % OCT_SUB;
            int code = s.voxel_code >= OCT_EDGE ? s.voxel_code - OCT_EDGE : s.voxel_code;
            double sg = (code & 1) ? 1.0 : -1.0;
            s.n_obj = {code < 2 ? sg : 0.0, (code == 2 || code == 3) ? sg : 0.0, code >= 4 ? sg : 0.0};
            if (s.voxel_code == OCT_VOLUME) s.n_obj = {1.0, 0.0, 0.0};  // hittable.rs:49-56
            s.pos_obj = oct_pos;
            s.voxel = sub / OCT_SUB;
            break;
        }
        default: break;  // ConstantMedium: normal (1,0,0), front_face true for the phase function
    }
    s.front_face = sh->kind == MRT_SHAPE_CONSTANT_MEDIUM ? true : (dot(sr.d, s.n_obj) <= 0.0);
}
template <class OP>
__device__ __forceinline__ D3 normal_to_world(OP ob, D3 n) {  // hittable.rs:215-220
    if (ob->identity) return n;  // (a pure translation too: its normal matrix is exactly the identity)
    // cgmath Matrix3 * Vector3 = col0*x + col1*y + col2*z; normal_m holds rows
    return {(ob->normal_m[0][0] * n.x + ob->normal_m[0][1] * n.y) + ob->normal_m[0][2] * n.z,
            (ob->normal_m[1][0] * n.x + ob->normal_m[1][1] * n.y) + ob->normal_m[1][2] * n.z,
            (ob->normal_m[2][0] * n.x + ob->normal_m[2][1] * n.y) + ob->normal_m[2][2] * n.z};
}
template <class OP>
__device__ __forceinline__ D3 point_to_world(OP ob, D3 p) {
    const int xf = ob->identity;
    if (xf == 1) return p;
    // (a pure translation: the products with the exact 0s and 1s of its linear part leave p + t, rounded once)
    if (xf == 2) return {p.x + ob->to_world.r[0][3], p.y + ob->to_world.r[1][3], p.z + ob->to_world.r[2][3]};
    return apply(&ob->to_world, p);
}

// ============================================================== textures (texture.rs)
// Perlin::noise / trilinear_interp / turbulence, texture.rs:95-160
template <class PN>
__device__ __noinline__ double perlin_noise(PN pn, D3 point) {
    double fx = floor(point.x), fy = floor(point.y), fz = floor(point.z);
    double u = point.x - fx, v = point.y - fy, w = point.z - fz;
    int i = f64_as_i32(fx), j = f64_as_i32(fy), k = f64_as_i32(fz);
    double uu = u * u * (3.0 - 2.0 * u), vv = v * v * (3.0 - 2.0 * v), ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                uint32_t idx = pn->perm_x[(i + di) & 255] ^ pn->perm_y[(j + dj) & 255] ^ pn->perm_z[(k + dk) & 255];
                const auto c = pn->ran_float[idx];
                double d = (c[0] * (u - (double)di) + c[1] * (v - (double)dj)) + c[2] * (w - (double)dk);  // cgmath dot
                accum += (((double)di * uu + (double)(1 - di) * (1.0 - uu)) * ((double)dj * vv + (double)(1 - dj) * (1.0 - vv))) *
                         ((double)dk * ww + (double)(1 - dk) * (1.0 - ww)) * d;
            }
    return accum;
}
template <class PN>
__device__ double perlin_turbulence(PN pn, D3 point, int depth) {
    double acum = 0.0, weight = 1.0;
    D3 tp = point;
#pragma nounroll
    for (int n = 0; n < depth; ++n) {
        acum += weight * perlin_noise(pn, tp);
        weight *= 0.5;
        tp = tp * 2.0;
    }
    return fabs(acum);
}

// The scene's first Perlin table staged in LDS by the kernels that have room for it (9 KB): a turbulence
// value is seven noise evaluations of 14 dependent table lookups each, and from LDS a lookup returns in a
// fraction of the time of a cache hit. `have` (not the pointer: LDS address 0 is a valid address) says
// whether the copy exists.
struct PerlinLds {
    const LDS_AS mrt_perlin* table;
    bool have;
};
__device__ __forceinline__ uint32_t f64_as_index(double v, uint32_t hi) {  // Rust `as usize` then clamp
    if (!(v > 0.0)) return 0;
    if (v >= (double)hi) return hi;
    return (uint32_t)v;
}
template <int FEAT>
__device__ F3 texture_color(SceneRef sv, const DObjRec* sh, const Surface& sf, PerlinLds pl = PerlinLds{nullptr, false}) {
    // SolidColor textures are folded into the object record
    if ((FEAT & FEAT_TEXTURE) == 0 || sh->tex_solid) return {sh->color[0], sh->color[1], sh->color[2]};
    const int tex = sh->texture;
    F3 acc{1.f, 1.f, 1.f};
    const D3 pos = sf.pos_obj;
    double u = 0.0, v = 0.0;
    bool have_uv = false;
    int stack[8];
    int sp = 0;
    int cur = tex;
    for (int guard = 0; guard < 64; ++guard) {
        const DTexture& t = sv.textures[cur];
        bool leaf = true;
        F3 c{0.f, 0.f, 0.f};
        switch (t.kind) {
            case MRT_TEX_SOLID: c = {t.color[0], t.color[1], t.color[2]}; break;
            case MRT_TEX_IMAGE: {  // parallel_image.rs:66-71
                if (!have_uv) surface_uv<FEAT>(sh, sf, u, v), have_uv = true;
                const DImage& im = sv.images[t.image];
                uint32_t x = f64_as_index(u * ((double)im.width - 1.0), (uint32_t)im.width - 1);
                double vv = 1.0 - v;
                uint32_t y = f64_as_index(vv * ((double)im.height - 1.0), (uint32_t)im.height - 1);
                const float* px = sv.image_pool + im.offset + ((size_t)x + (size_t)y * (size_t)im.width) * 3;
                c = {px[0], px[1], px[2]};
                break;
            }
            case MRT_TEX_DEBUG_V:  // texture.rs:231-238
                if (!have_uv) surface_uv<FEAT>(sh, sf, u, v), have_uv = true;
                c = (v != v) ? F3{0.f, 0.f, 0.f} : F3{(float)v, (float)v, (float)v};
                break;
            case MRT_TEX_PERLIN: {  // texture.rs:185-189
                if constexpr ((FEAT & FEAT_PERLIN) != 0) {
                    float f = pl.have && t.image == 0 ? (float)perlin_turbulence(pl.table, 2.0 * pos, 7)
                                                      : (float)perlin_turbulence(sv.perlins + t.image, 2.0 * pos, 7);
                    c = {f, f, f};
                }
                break;
            }
            case MRT_TEX_MULTIPLY:  // texture.rs:29-31
                if (sp < 8) stack[sp++] = t.b;
                cur = t.a;
                leaf = false;
                break;
            default: {  // MRT_TEX_CHECKER, texture.rs:62-69
                double s = sin(10.0 * pos.x) * sin(10.0 * pos.y) * sin(10.0 * pos.z);
                cur = s < 0.0 ? t.a : t.b;
                leaf = false;
                break;
            }
        }
        if (leaf) {
            acc = acc * c;
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    return acc;
}

// ============================================================== light list (pdf.rs, world.rs:98-110)
// prob() of the light shapes (rect.rs:94-106,214-227,339-351; sphere.rs:63-70; render_box.rs:137-155)
template <int AXIS>
__device__ __forceinline__ double rect_prob(double a0, double a1, double b0, double b1, double k, D3 origin) {
    constexpr int IA = AXIS == 0 ? 1 : 0;
    constexpr int IB = AXIS == 2 ? 1 : 2;
    double c[3];
    c[IA] = 0.5 * (a0 + a1);
    c[IB] = 0.5 * (b0 + b1);
    c[AXIS] = 0.5 * (k + k);
    D3 to_light = D3{c[0], c[1], c[2]} - origin;
    double cos_alpha = fabs(comp(normalize(to_light), AXIS));
    if (cos_alpha < 0.00001) return 0.0;
    double area = (a1 - a0) * (b1 - b0);
    return dot(to_light, to_light) * rcp_fast(cos_alpha * area);
}
template <int FEAT, class OP>
__device__ double shape_prob(OP sh, const RayD& r) {
    auto p = sh->p;
    switch (sh->gkind) {  // a medium answers for its boundary (constant_medium.rs:98-100)
        case MRT_SHAPE_SPHERE: {
            if constexpr ((FEAT & FEAT_SPHERE) != 0) {
                double area = M_PI * (p[0] * p[0]);
                D3 to_light = D3{p[1], p[2], p[3]} - r.o;
                return dot(to_light, to_light) / area;
            }
            return 0.0;
        }
        case MRT_SHAPE_XY_RECT: return rect_prob<2>(p[0], p[1], p[2], p[3], p[4], r.o);
        case MRT_SHAPE_XZ_RECT: return rect_prob<1>(p[0], p[1], p[2], p[3], p[4], r.o);
        case MRT_SHAPE_YZ_RECT: return rect_prob<0>(p[0], p[1], p[2], p[3], p[4], r.o);
        case MRT_SHAPE_RENDER_BOX: {
            double area = 0.0;
            area += r.d.x >= 0.0 ? rect_prob<0>(p[1], p[4], p[2], p[5], p[0], r.o) : rect_prob<0>(p[1], p[4], p[2], p[5], p[3], r.o);
            area += r.d.y >= 0.0 ? rect_prob<1>(p[0], p[3], p[2], p[5], p[1], r.o) : rect_prob<1>(p[0], p[3], p[2], p[5], p[4], r.o);
            area += r.d.z >= 0.0 ? rect_prob<2>(p[0], p[3], p[1], p[4], p[2], r.o) : rect_prob<2>(p[0], p[3], p[1], p[4], p[5], r.o);
            return area;
        }
        default: return 0.0;
    }
}
struct AreaInfo {  // hittable.rs:450-456
    RayD to_area;
    double area;
    D3 direction, normal, end_point;
};
template <int AXIS>
__device__ __forceinline__ void rect_generate(double a0, double a1, double b0, double b1, double k, double flip,
                                              D3 origin, double time, double ua, double ub, AreaInfo& o) {
    constexpr int IA = AXIS == 0 ? 1 : 0;
    constexpr int IB = AXIS == 2 ? 1 : 2;
    double c[3];
    c[IA] = ua * (a1 - a0) + a0;  // rand_scalar(a0, a1) on the uniform of MRT_DIM_LIGHT_A
    c[IB] = ub * (b1 - b0) + b0;  // rand_scalar(b0, b1) on the uniform of MRT_DIM_LIGHT_B
    c[AXIS] = k;
    D3 end_point{c[0], c[1], c[2]};
    o.to_area = {origin, normalize(end_point - origin), time};
    o.normal = {AXIS == 0 ? flip : 0.0, AXIS == 1 ? flip : 0.0, AXIS == 2 ? flip : 0.0};
    o.area = (a1 - a0) * (b1 - b0);
    o.direction = end_point - origin;
    o.end_point = end_point;
}
template <int FEAT, class OP>
__device__ void shape_generate(OP sh, D3 origin, double time, const Rng& g, double ua, double ub, AreaInfo& o,
                               const D3* unit_vec = nullptr) {
    auto p = sh->p;
    switch (sh->gkind) {  // a medium answers for its boundary (constant_medium.rs:101-103)
        case MRT_SHAPE_SPHERE: {  // sphere.rs:72-87
            if constexpr ((FEAT & FEAT_SPHERE) != 0) {
                // (`unit_vec`: the same vector, when the caller had the wave work it out)
                D3 sd = unit_vec ? *unit_vec : rand_unit_vec(g, MRT_DIM_LIGHT_SPHERE);
                D3 end_point = D3{p[1], p[2], p[3]} + p[0] * sd;
                o.to_area = {origin, normalize(end_point - origin), time};
                o.normal = sd;
                o.area = M_PI * (p[0] * p[0]);
                o.direction = end_point - origin;
                o.end_point = end_point;
            }
            break;
        }
        case MRT_SHAPE_XY_RECT: rect_generate<2>(p[0], p[1], p[2], p[3], p[4], p[5], origin, time, ua, ub, o); break;
        case MRT_SHAPE_XZ_RECT: rect_generate<1>(p[0], p[1], p[2], p[3], p[4], p[5], origin, time, ua, ub, o); break;
        case MRT_SHAPE_YZ_RECT: rect_generate<0>(p[0], p[1], p[2], p[3], p[4], p[5], origin, time, ua, ub, o); break;
        default: {  // RenderBox, render_box.rs:157-181
            uint32_t face = rand_u32(g, 0, 3, MRT_DIM_LIGHT_FACE);
            D3 c{(p[0] + p[3]) / 2.0, (p[1] + p[4]) / 2.0, (p[2] + p[5]) / 2.0};
            D3 to_self = c - origin;
            if (face == 0) {
                if (to_self.x >= 0.0) rect_generate<0>(p[1], p[4], p[2], p[5], p[0], -1.0, origin, time, ua, ub, o);
                else rect_generate<0>(p[1], p[4], p[2], p[5], p[3], 1.0, origin, time, ua, ub, o);
            } else if (face == 1) {
                if (to_self.y >= 0.0) rect_generate<1>(p[0], p[3], p[2], p[5], p[1], -1.0, origin, time, ua, ub, o);
                else rect_generate<1>(p[0], p[3], p[2], p[5], p[4], 1.0, origin, time, ua, ub, o);
            } else {
                if (to_self.z >= 0.0) rect_generate<2>(p[0], p[3], p[1], p[4], p[2], -1.0, origin, time, ua, ub, o);
                else rect_generate<2>(p[0], p[3], p[1], p[4], p[5], 1.0, origin, time, ua, ub, o);
            }
            break;
        }
    }
}
// Object::generate_ray_in_area, hittable.rs:289-312
template <int FEAT, class OP>
__device__ void object_generate(OP ob, D3 origin, double time, const Rng& g, double ua, double ub, AreaInfo& out,
                                const D3* unit_vec = nullptr) {
    if (ob->identity == 1) {
        shape_generate<FEAT>(ob, origin, time, g, ua, ub, out, unit_vec);
        // the reference still routes every vector through the (identity) inverse
        D3 o = out.to_area.o;
        out.to_area.d = (o + out.to_area.d) - o;
        out.direction = (o + out.direction) - o;
        out.normal = (o + out.normal) - o;
        return;
    }
    AreaInfo a;
    shape_generate<FEAT>(ob, apply(&ob->to_object, origin), time, g, ua, ub, a, unit_vec);
    auto inv = &ob->to_world;
    out.end_point = apply(inv, a.end_point);
    out.to_area = xform_ray(inv, false, a.to_area);
    D3 sc = apply(inv, D3{1.0, 0.0, 0.0}) - apply(inv, D3{0.0, 0.0, 0.0});
    out.area = a.area * fabs(magnitude(sc));
    out.direction = apply(inv, a.to_area.o + a.direction) - out.to_area.o;
    out.normal = apply(inv, a.to_area.o + a.normal) - out.to_area.o;
}

// LightPdf::value, pdf.rs:82-99 with World::nearest_light_hit, world.rs:98-110. Returns false for None.
template <int FEAT>
__device__ bool light_pdf_value(SceneRef sv, const RayD& ray, const Rng& g, uint32_t& light_queries,
                                double& value) {
    constexpr int LF = FEAT & ~(FEAT_OCTREE | FEAT_VOLUME);  // an octree is never a light
    light_queries++;
    bool any = false;
    double best_t = 0.0;
    int best = -1, best_sub = 0;
    uint32_t ordinal = 0;
    auto lights = as_const(sv.lights);  // the loop index is wave-uniform: scalar loads
    D3 inv_d{rcp_fast(ray.d.x), rcp_fast(ray.d.y), rcp_fast(ray.d.z)};
    for (int i = 0; i < sv.n_lights; ++i) {
        auto lob = lights + i;
        double t;
        int sub;
        D3 unused_pos;
        uint32_t no_octree = 0;
        bool h = object_hit<LF>(sv, lob, ray, inv_d, ray.time, DBL_MAX, g, MRT_DIM_LIGHT_QUERY_MEDIUM + ordinal, no_octree, t, sub, unused_pos);
        if constexpr ((LF & FEAT_MEDIUM) != 0) {
            if (lob->kind == MRT_SHAPE_CONSTANT_MEDIUM) ordinal++;
        }
        if (!h) continue;
        if (!any || !(best_t < t)) {  // reduce(|acc, x| if acc.t < x.t { acc } else { x })
            any = true, best_t = t, best = i, best_sub = sub;
        }
    }
    if (!any) return false;
    // (a one-entry list: the light's record is the same for every lane and is read with scalar loads)
    auto value_at = [&](auto ob) {
        Surface s;
        surface_at<LF>(ob, ray, best_t, best_sub, D3{0.0, 0.0, 0.0}, s, false);
        D3 hit_pos = point_to_world(ob, s.pos_obj);
        D3 hit_n = normal_to_world(ob, s.n_obj);
        D3 to_light = hit_pos - ray.o;
        double light_cos = fabs(dot(normalize(to_light), hit_n));
        if (light_cos >= 0.000001) {
            // Object::prob applies the INVERSE transform to the ray (hittable.rs:284-287)
            RayD pr{ray.o, normalize(to_light), best_t};
            RayD tr = xform_ray(&ob->to_world, ob->identity == 1, pr);
            value = shape_prob<LF>(ob, tr);
        } else {
            value = 0.0;
        }
    };
    if (sv.n_lights == 1) value_at(as_const(sv.lights));
    else value_at(sv.lights + best);
    return true;
}

// prelude.rs:103-130
struct Onb {
    D3 u, v, w;
};
__device__ __forceinline__ Onb onb_from_w(D3 n) {
    // Normals of rectangles, box faces and voxels are unit axis vectors: w is n itself, the cross product with
    // the helper axis is again a unit axis vector, and both normalisations multiply by exactly 1 -- which is
    // what the reference computes (cgmath normalize, v * (1 / 1)). When every lane here has such a normal the
    // two reciprocal square roots are left out.
    const bool unit_axis = (n.x == 0.0 && n.y == 0.0 && fabs(n.z) == 1.0) || (n.x == 0.0 && fabs(n.y) == 1.0 && n.z == 0.0) ||
                           (fabs(n.x) == 1.0 && n.y == 0.0 && n.z == 0.0);
    if (__ballot(!unit_axis) == 0) {
        D3 a = fabs(n.x) > 0.9 ? D3{0.0, 1.0, 0.0} : D3{1.0, 0.0, 0.0};
        D3 v = cross(n, a);
        D3 u = cross(n, v);
        return {u, v, n};
    }
    D3 w = normalize(n);
    D3 a = fabs(w.x) > 0.9 ? D3{0.0, 1.0, 0.0} : D3{1.0, 0.0, 0.0};
    D3 v = normalize(cross(w, a));
    D3 u = cross(w, v);
    return {u, v, w};
}

// LambertianPDF::generate, pdf.rs:234-282, in two steps around the wave's unit-vector sampler: the choice of
// the pdf with the draws every choice starts with (and whether it will need a vector in the unit ball), then
// the sample itself.
struct LambertChoice {
    uint32_t r;      // 0 cosine lobe, 1 light list, 2 sky
    uint32_t light;  // index into the light list (r == 1)
    double ua, ub;
};
template <int FEAT>
__device__ __forceinline__ void lambertian_choose(SceneRef sv, const Rng& g, LambertChoice& c, bool& want_vec,
                                                  uint32_t& vec_dim) {
    const uint32_t r = rand_u32(g, 0, 3, MRT_DIM_PDF_CHOICE);
    // Every branch starts with two uniforms (cosine lobe: r1, r2; light: the two coordinates on a
    // flat light; sky: the sun sample). They are drawn here for all lanes at once, each lane at the
    // dimensions of its own branch, instead of three times for a third of the lanes.
    bool has_sun = false;
    if constexpr ((FEAT & FEAT_SKY) != 0) has_sun = sv.sun.has_sun != 0;
    const uint64_t add_a = r == 0 ? rng_dim_addend(MRT_DIM_COS_R1)
                                  : (r == 1 ? rng_dim_addend(MRT_DIM_LIGHT_A) : rng_dim_addend(MRT_DIM_SUN_R));
    const uint64_t add_b = r == 0 ? rng_dim_addend(MRT_DIM_COS_R2)
                                  : (r == 1 ? rng_dim_addend(MRT_DIM_LIGHT_B) : rng_dim_addend(MRT_DIM_SUN_THETA));
    c.r = r;
    c.ua = rng_f64_addend(g, add_a), c.ub = rng_f64_addend(g, add_b);
    c.light = 0;
    if (r == 1) {
        // a one-entry list needs no draw (x % 1 == 0; draws are addressed by dimension)
        if (sv.n_lights > 1) c.light = rand_u32(g, 0, (uint32_t)sv.n_lights, MRT_DIM_LIGHT_INDEX);
        if constexpr ((FEAT & FEAT_SPHERE) != 0) {
            const int gkind = sv.n_lights == 1 ? as_const(sv.lights)->gkind : (sv.n_lights > 0 ? sv.lights[c.light].gkind : -1);
            if (gkind == MRT_SHAPE_SPHERE) want_vec = true, vec_dim = MRT_DIM_LIGHT_SPHERE;
        }
    } else if (r == 2 && !has_sun) {
        want_vec = true, vec_dim = MRT_DIM_SKY;
    }
}
// Returns false for None. `unit_vec`: what rand_unit_vec gives at the dimensions lambertian_choose asked for.
template <int FEAT>
__device__ bool lambertian_generate(SceneRef sv, D3 pdf_normal, double incoming_time, D3 hit_point,
                                    const Rng& g, const LambertChoice& c, D3 unit_vec, uint32_t& light_queries, D3& out_dir,
                                    double& out_value, Stamps& st) {
    REGION_BEGIN(gen);
    Onb uvw = onb_from_w(pdf_normal);
    const uint32_t r = c.r;
    const double ua = c.ua, ub = c.ub;
    bool has_sun = false;
    if constexpr ((FEAT & FEAT_SKY) != 0) has_sun = sv.sun.has_sun != 0;
    D3 dir;
    double own;
    if (r == 0) {  // CosinePdf::generate, pdf.rs:45-60; random_cosine_direction, prelude.rs:145-153
        double r1 = ua, r2 = ub;  // rand_scalar(0, 1) is the uniform itself
        double z = sqrt(1.0 - r2);
        double sr2 = sqrt(r2);
        double sphi, cphi;
        sincospi(2.0 * r1, &sphi, &cphi);  // phi = 2 pi r1
        double x = cphi * sr2, y = sphi * sr2;
        dir = normalize(x * uvw.u + y * uvw.v + z * uvw.w);
        double c = dot(dir, uvw.w);
        own = c <= 0.0 ? 0.0 : c * INV_PI;
        REGION_END(st, ST_GEN_COS, gen);
    } else if (r == 1) {  // LightPdf::generate, pdf.rs:104-130
        if (sv.n_lights == 0) return false;
        AreaInfo ai;
        constexpr int LF = FEAT & ~(FEAT_OCTREE | FEAT_VOLUME);
        // (a one-entry list: the light's record is read with scalar loads)
        if (sv.n_lights == 1) object_generate<LF>(as_const(sv.lights), hit_point, incoming_time, g, ua, ub, ai, &unit_vec);
        else object_generate<LF>(sv.lights + c.light, hit_point, incoming_time, g, ua, ub, ai, &unit_vec);
        D3 nd = normalize(ai.to_area.d);
        double light_cos = fabs(dot(nd, ai.normal));
        double dist_squared = dot(ai.direction, ai.direction);
        if (!(light_cos >= 0.000001)) return false;
        own = dist_squared * rcp_fast(light_cos * ai.area);
        dir = nd;
        REGION_END(st, ST_GEN_LIGHT, gen);
    } else {  // SkyPdf::generate, pdf.rs:154-193
        if (!has_sun) {
            dir = unit_vec;
            own = 4.0 * M_PI;
        } else if constexpr ((FEAT & FEAT_SKY) != 0) {
            double rr = sqrt(ua);                              // rand_scalar(0, 1)
            double theta = ub * (2.0 * M_PI - 0.0) + 0.0;      // rand_scalar(0, 2 pi)
            rr = rr * sv.sun.radius;
            D3 svv = ld3(sv.sun_dir);
            D3 k = normalize(cross(svv, D3{0.0, 0.0, 1.0}));
            dir = cos(rr) * svv + cross(k, svv) * sin(theta) + k * dot(k, svv) * (1.0 - cos(theta));
            double area = M_PI * (sv.sun.radius * sv.sun.radius);
            own = area / (4.0 * M_PI);
        }
        REGION_END(st, ST_GEN_SKY, gen);
    }
    REGION_END(st, ST_LAMBERT_GEN, gen);
    RayD value_ray{hit_point, dir, 0.0};
    double sum = 0.0;
    int total = 0;
    if (r != 1) {  // light_pdf.value
        REGION_BEGIN(lv);
        double lv;
        if (light_pdf_value<FEAT>(sv, value_ray, g, light_queries, lv)) sum += lv, total++;
        REGION_END(st, ST_LIGHT_VALUE, lv);
    }
    if (r != 0) {  // sin_pdf.value, pdf.rs:32-39 (direction NOT normalised here)
        double c = dot(dir, uvw.w);
        sum += c <= 0.0 ? 0.0 : c * INV_PI;
        total++;
    }
    if (r != 2) {  // sky_pdf.value, pdf.rs:135-148
        double svl = 1.0;
        if constexpr ((FEAT & FEAT_SKY) != 0) {
            if (sv.sun.has_sun) svl = dot(dir, ld3(sv.sun_dir)) >= sv.sun_cos_radius ? 1.0 : 0.0;
        }
        sum += svl;
        total++;
    }
    if (total == 0) return false;
    out_dir = dir;
    out_value = (sum + own) / (double)(total + 1);
    return true;
}

// ============================================================== backgrounds
__device__ __forceinline__ F3 sky_color(double intensity, const RayD& ray) {  // sky.rs:13-28
    D3 unit = normalize(ray.d);
    double t = 0.5 * (unit.y + 1.0);
    float a = (float)(1.0 - t), b = (float)t, k = (float)intensity;
    F3 c{1.0f * a + 0.5f * b, 1.0f * a + 0.7f * b, 1.0f * a + 1.0f * b};
    return {c.r * k, c.g * k, c.b * k};
}
template <int FEAT>
__device__ F3 background_color(SceneRef sv, const RayD& ray) {
    const auto& b = sv.bg;
    if ((FEAT & FEAT_SKY) == 0 || b.kind == MRT_BG_CONSTANT) return {b.color[0], b.color[1], b.color[2]};  // constant_color.rs:12-14
    if (b.kind == MRT_BG_SUN_SKY) {  // sun_sky.rs:32-60
        // (sun_ray = (cos phi cos theta, sin phi, cos phi sin theta) and cos(radius) come from the host)
        double sun_cos = dot(ld3(sv.bg_sun_ray), normalize(ray.d));
        if (sun_cos > sv.bg_cos_sun_radius && sun_cos > 0.0) {
            float k = (float)b.sun_brightness;
            return {k, k, k};
        }
    }
    return sky_color(b.intensity, ray);
}

// ============================================================== camera (camera.rs:88-106, ray_tracer.rs:444-447)
// a / d by the reciprocal the host gives (div_by_reciprocal); inv = NaN (d is out of its range): the division.
__device__ __forceinline__ double div_by(double a, double d, double inv) {
    if (!(inv == inv)) return a / d;  // (the same for every lane)
    return div_by_reciprocal(a, d, inv);
}
__host__ __device__ inline double reciprocal_for_div_by(double d) {
    return (d >= 1.0 && d <= 1e15) ? 1.0 / d : __builtin_nan("");
}
// `inv_w1`, `inv_h1`: reciprocal_for_div_by(width - 1), (height - 1)
// Camera::get_ray (camera.rs:88-98) without the lens sample: the ray through the lens centre.
__device__ __forceinline__ RayD camera_ray_pinhole(SceneRef sv, uint32_t x, uint32_t y, uint32_t width, uint32_t height,
                                                   const Rng& g, double inv_w1, double inv_h1) {
    const auto& c = sv.cam;
    double s = div_by((double)x + rand_scalar(g, 0.0, 1.0, MRT_DIM_JITTER_U), (double)width - 1.0, inv_w1);
    double t = div_by((double)y + rand_scalar(g, 0.0, 1.0, MRT_DIM_JITTER_V), (double)height - 1.0, inv_h1);
    RayD r;
    r.o = ld3(c.origin);
    r.d = ld3(c.lower_left_corner) + s * ld3(c.horizontal) + t * ld3(c.vertical) - ld3(c.origin);
    // rand_scalar(start, end) is start + u * 0 when the shutter interval is empty
    r.time = c.start_time == c.end_time ? c.start_time : rand_scalar(g, c.start_time, c.end_time, MRT_DIM_TIME);
    return r;
}
// The lens sample `p` (a point in the unit disk) moves the direction: rd = lens_radius * p, offset = u * rd.x + v * rd.y,
// direction - offset (camera.rs:91-96; the origin stays the camera's).
__device__ __forceinline__ void camera_lens(SceneRef sv, D3 p, RayD& r) {
    const auto& c = sv.cam;
    D3 rd = c.lens_radius * p;
    D3 offset = ld3(c.u) * rd.x + ld3(c.v) * rd.y;
    r.d = r.d - offset;
}
// The whole of get_ray for one lane (the probes and the single-bounce shaders; the segment kernel samples the
// lens for all its lanes at once, wave_rand_reject<2>).
__device__ RayD camera_ray(SceneRef sv, uint32_t x, uint32_t y, uint32_t width, uint32_t height,
                           const Rng& g, double inv_w1, double inv_h1) {
    RayD r = camera_ray_pinhole(sv, x, y, width, height, g, inv_w1, inv_h1);
    // rand_in_unit_disk (camera.rs:99-106). With a zero lens radius the sample is multiplied by 0, the offset is a
    // zero vector and subtracting it changes nothing; the draws are addressed by dimension, so skipping them
    // changes nothing else.
    if (sv.cam.lens_radius != 0.0) {
        D3 p{0.0, 0.0, 0.0};
#pragma nounroll
        for (uint32_t k = 0; k < MRT_MAX_REJECT; ++k) {
            D3 q{rand_scalar(g, -1.0, 1.0, MRT_DIM_LENS + 2 * k), rand_scalar(g, -1.0, 1.0, MRT_DIM_LENS + 2 * k + 1), 0.0};
            if (dot(q, q) < 1.0) {
                p = q;
                break;
            }
        }
        camera_lens(sv, p, r);
    }
    return r;
}

// lib.rs:8-10
__device__ __forceinline__ D3 reflect(D3 v, D3 n) { return v - 2.0 * dot(v, n) * n; }

// ============================================================== one path segment
// RayTracingShader::ray_color, ray_tracer.rs:159-303, unrolled from recursion into
// "throughput *= attenuation [/ value]"; the terminal value is emitted light, background or black.
// Returns true when the path continues with `ray` / `thr` updated; otherwise `radiance` holds the
// path's contribution (already multiplied by the throughput).
// Material of a voxel hit (VoxelMaterial, fast_oct_tree/voxel.rs:57-62) in terms of the MRT_MAT_* kinds:
// Lambertian -> LAMBERTIAN, Reflect -> METAL with the voxel's fuzz, Volume -> MAT_VOXEL_VOLUME.
constexpr int MAT_VOXEL_VOLUME = 100;
template <int FEAT>
__device__ __forceinline__ int voxel_material(SceneRef sv, const Surface& s, F3& att, double& fuzz) {
    const mrt_voxel& vx = sv.oct_voxels[s.voxel];
    att = {vx.albedo[0], vx.albedo[1], vx.albedo[2]};
    fuzz = 0.0;
    if constexpr ((FEAT & FEAT_VOLUME) != 0) {
        if (s.voxel_code == OCT_VOLUME) return MAT_VOXEL_VOLUME;  // VolumeVoxel::volume_material
        if (s.voxel_code >= OCT_EDGE) {                            // solid_material.to_material() of the edge effect
            att = {vx.edge_albedo[0], vx.edge_albedo[1], vx.edge_albedo[2]};
            fuzz = (double)vx.edge_fuzz;
            return vx.edge_kind == MRT_VOX_EDGE_REFLECT ? (int)MRT_MAT_METAL : (int)MRT_MAT_LAMBERTIAN;
        }
        if (vx.kind == MRT_VOX_REFLECT) {
            fuzz = (double)vx.fuzz;
            return MRT_MAT_METAL;
        }
    }
    return MRT_MAT_LAMBERTIAN;
}

// Material response at the nearest hit `h` (or the background when `hit` is false), in two steps around
// the wave's unit-vector sampler (wave_rand_unit_vec): shade_prepare ends the paths that end here (background,
// light, absorbing metal) and says whether the response needs a vector in the unit ball and at which
// dimensions; shade_finish, for the lanes shade_prepare returned true for, scatters.
struct ShadeCtx {
    const DObjRec* ob;
    Surface s;
    int mkind;
    double fuzz;
    F3 att;
    LambertChoice lc;
};
template <int FEAT>
__device__ bool shade_prepare(SceneRef sv, const RayD& ray, const F3& thr, const Rng& g, const HitD& h, bool hit,
                              F3& radiance, ShadeCtx& c, bool& want_vec, uint32_t& vec_dim, Stamps& st, PerlinLds pl) {
    radiance = {0.f, 0.f, 0.f};
    if (!hit) {
        F3 bg = background_color<FEAT>(sv, ray);
        radiance = thr * bg;
        return false;
    }
    REGION_BEGIN(surf);
    const DObjRec* ob = sv.objects + h.obj;
    c.ob = ob;
    Surface& s = c.s;
    surface_at<FEAT>(ob, ray, h.t, h.sub, h.pos, s);
    const bool voxel_mat = (FEAT & FEAT_OCTREE) != 0 && s.voxel >= 0;
    int mkind = ob->mat_kind;
    double fuzz = ob->fuzz;
    // Texture value of the material (emitted colour / albedo); evaluated once for every material kind.
    F3 att{ob->color[0], ob->color[1], ob->color[2]};
    if (voxel_mat) {
        if constexpr ((FEAT & FEAT_OCTREE) != 0) mkind = voxel_material<FEAT>(sv, s, att, fuzz);
    } else if (mkind != MRT_MAT_DIELECTRIC) {
        att = texture_color<FEAT>(sv, ob, s, pl);
    }
    c.mkind = mkind, c.fuzz = fuzz, c.att = att;
    REGION_END(st, ST_SURFACE, surf);
    // HitRecord::new (hittable.rs:414-448): emit first, else scatter, else nothing.
    switch (mkind) {
        case MRT_MAT_DIFFUSE_LIGHT: {  // material.rs:236-248
            if (s.front_face) radiance = thr * att;
            return false;
        }
        case MAT_VOXEL_VOLUME:  // voxel.rs:90-95: IsotropicPdf (pdf.rs:63-79) with the Lambertian scattering pdf
            if constexpr ((FEAT & FEAT_VOLUME) != 0) {
                want_vec = true, vec_dim = MRT_DIM_ISOTROPIC_PDF;
                return true;
            }
            return false;
        case MRT_MAT_LAMBERTIAN:  // material.rs:54-64 + ray_tracer.rs:230-268
            lambertian_choose<FEAT>(sv, g, c.lc, want_vec, vec_dim);
            return true;
        case MRT_MAT_METAL: {  // material.rs:105-124; also VoxelMaterial::Reflect, voxel.rs:96-115
            if constexpr ((FEAT & (FEAT_SPECULAR | FEAT_VOLUME)) != 0) {
                D3 reflected = reflect(normalize(s.dir_obj), s.n_obj);
                if (!(dot(reflected, s.n_obj) > 0.0)) return false;
                want_vec = true, vec_dim = MRT_DIM_METAL_FUZZ;
                return true;
            }
            return false;
        }
        case MRT_MAT_DIELECTRIC: return (FEAT & FEAT_SPECULAR) != 0;
        default:  // MRT_MAT_ISOTROPIC, material.rs:274-285
            want_vec = true, vec_dim = MRT_DIM_ISOTROPIC;
            return true;
    }
}
// `unit_vec`: rand_unit_vec at the dimensions shade_prepare asked for. `t` = the hit's t.
template <int FEAT>
__device__ bool shade_finish(SceneRef sv, RayD& ray, F3& thr, const Rng& g, double t, const ShadeCtx& c, D3 unit_vec,
                             uint32_t& light_queries, Stamps& st, float* cap_out = nullptr) {
    const DObjRec* ob = c.ob;
    const Surface& s = c.s;
    const F3 att = c.att;
    switch (c.mkind) {
        case MAT_VOXEL_VOLUME:
        case MRT_MAT_LAMBERTIAN: {
            D3 pos_w = point_to_world(ob, s.pos_obj);
            D3 n_w = normal_to_world(ob, s.n_obj);
            D3 dir;
            double value;
            if ((FEAT & FEAT_VOLUME) != 0 && c.mkind == MAT_VOXEL_VOLUME) {
                dir = unit_vec;
                value = 1.0;
            } else if (!lambertian_generate<FEAT>(sv, s.n_obj, ray.time, pos_w, g, c.lc, unit_vec, light_queries, dir, value, st)) {
                return false;
            }
            double cosine = dot(n_w, normalize(dir));  // Lambertian::scattering_pdf_fn, material.rs:30-41
            if (cosine < 0.0) return false;
            double scattering_pdf = cosine * INV_PI;
            if (scattering_pdf == 0.0) return false;
            float fv = (float)(value * rcp_fast(scattering_pdf));
            thr = {thr.r * att.r / fv, thr.g * att.g / fv, thr.b * att.b / fv};
            ray = {pos_w, dir, t};
            break;
        }
        case MRT_MAT_METAL: {  // the specular ray stays in OBJECT space
            if constexpr ((FEAT & (FEAT_SPECULAR | FEAT_VOLUME)) != 0) {
                D3 reflected = reflect(normalize(s.dir_obj), s.n_obj);
                thr = thr * att;
                ray = {s.pos_obj, reflected + c.fuzz * unit_vec, s.time};
                break;
            }
            return false;
        }
        case MRT_MAT_DIELECTRIC: {  // material.rs:141-196
            if constexpr ((FEAT & FEAT_SPECULAR) != 0) {
                double ratio = s.front_face ? 1.0 / ob->index_refraction : ob->index_refraction;
                D3 unit = normalize(s.dir_obj);
                double cos_theta = fmin(dot(s.n_obj, -1.0 * unit), 1.0);
                double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
                bool can_not_refract = (ratio * sin_theta) > 1.0;
                double r0 = (1.0 - ratio) / (1.0 + ratio);
                r0 = r0 * r0;
                double om = 1.0 - cos_theta;
                double refl = r0 + (1.0 - r0) * (om * om * om * om * om);
                D3 dir;
                if (can_not_refract || refl > rng_f64(g, MRT_DIM_DIELECTRIC)) {
                    dir = reflect(unit, s.n_obj);
                } else {
                    double ct = fmin(dot(s.n_obj, -1.0 * unit), 1.0);
                    D3 perp = ratio * (unit + ct * s.n_obj);
                    D3 par = -1.0 * s.n_obj * sqrt(fabs(1.0 - dot(perp, perp)));
                    dir = perp + par;
                }
                thr = thr * att;
                ray = {s.pos_obj, dir, s.time};
                break;
            }
            return false;
        }
        default: {  // MRT_MAT_ISOTROPIC
            thr = thr * att;
            ray = {s.pos_obj, unit_vec, s.time};
            break;
        }
    }
    // A path whose throughput is exactly zero can only ever add 0 (the reference still walks it).
    if (!sv.trace_zero_throughput && thr.r == 0.f && thr.g == 0.f && thr.b == 0.f) return false;
    if constexpr ((FEAT & FEAT_SPHERE) != 0) {
        // (kernels with the candidate search: what the next query can be told about its answer; the new ray of an
        // untransformed sphere is in world space whatever the material)
        if (cap_out && ob->kind == MRT_SHAPE_SPHERE && ob->identity == 1)
            *cap_out = sphere_far_root_cap(D3{ob->p[1], ob->p[2], ob->p[3]}, ob->p[0], ray);
    }
    return true;
}

// One segment for one lane: nearest hit + material response. `live` marks the lanes that carry a
// path; every lane of the wave must call this (the wave-synchronous traversal and the unit-vector
// sampler are wave-wide).
// `walking` / `may_suspend`: see nearest_hit_wave. A lane whose walk is suspended has no answer in this turn: nothing
// of its path changes, and the caller calls again with `walking` as it came back.
template <bool WAVE, int FEAT>
__device__ bool path_segment(SceneRef sv, RayD& ray, F3& thr, const Rng& g, bool live, F3& radiance,
                             uint32_t& rays, uint32_t& light_queries, Stamps& st, CandLds cand_bits, PerlinLds pl, bool& walking,
                             bool may_suspend, float& next_cap) {
    radiance = {0.f, 0.f, 0.f};
    HitD h{0.0, -1, 0, {0.0, 0.0, 0.0}};
    bool hit;
    REGION_BEGIN(trav);
    if (WAVE) {
        if constexpr ((FEAT & FEAT_OCTREE) != 0 && (FEAT & FEAT_VOLUME) == 0) {
            hit = nearest_hit_wave<FEAT>(sv, ray, g, live, h, st, &walking, may_suspend);
            live = live && !walking;
        } else {
            hit = nearest_hit_wave<FEAT>(sv, ray, g, live, h, st);
        }
    } else {
        bool again = false;
        hit = nearest_hit_lane<FEAT>(sv, ray, g, live, h, st, cand_bits, next_cap, &again);
        next_cap = INFINITY;
        walking = again;  // (no answer in this turn: nothing of the path moves, the next turn asks without a cap)
        live = live && !again;
    }
    REGION_END(st, ST_TRAVERSE, trav);
    REGION_BEGIN(shade);
    ShadeCtx c;
    bool more = false, want_vec = false, cont = false;
    uint32_t vec_dim = 0;
    if (live) {
        rays++;
        more = shade_prepare<FEAT>(sv, ray, thr, g, h, hit, radiance, c, want_vec, vec_dim, st, pl);
    }
    REGION_BEGIN(uvec);
    const D3 unit_vec = wave_rand_unit_vec(want_vec, g.base, vec_dim);
    REGION_END(st, ST_UNIT_VEC, uvec);
    if (more) cont = shade_finish<FEAT>(sv, ray, thr, g, h.t, c, unit_vec, light_queries, st, WAVE ? nullptr : &next_cap);
    REGION_END(st, ST_SHADE, shade);
    return cont;
}

// ============================================================== kernels
// This lane's bit column and this wave's queue of the workgroup's candidate-search arrays, bits cleared.
__device__ __forceinline__ CandLds candidate_lds(SceneRef sv) {
    const uint32_t tpb = blockDim.x;
    uint32_t* base = dynamic_lds();
    CandLds lds{(LDS_AS uint32_t*)base + threadIdx.x,
                (LDS_AS uint16_t*)(base + (CAND_WORDS + 1) * tpb) + (threadIdx.x >> 6) * CAND_QUEUE, tpb};
    for (int w = 0; w < sv.n_cand_words; ++w) lds.bits[w * tpb] = 0;
    lds.bits[CAND_WORDS * tpb] = 0;
    return lds;
}
// Waves per SIMD the segment kernel is compiled for (512 / waves registers per lane), see seg_waves.
// Threads per workgroup of the segment kernel: single waves for the wave-synchronous kernels (nothing
// is shared between the waves of a workgroup, and a wave's slot is free again the moment it ends:
// +4 % on Cornell smoke, +8 % on the voxel city over 256), 256 for the per-lane kernels (measured
// better with the candidate search's LDS traffic).
constexpr int seg_block(bool wave) { return wave ? 64 : 256; }
static_assert(seg_block(true) == 64, "octree_first_leaf takes the wave-synchronous kernels' workgroup for 64 threads");
// Dynamic LDS a kernel of `threads` threads needs: the octree ancestor cache where octree code
// exists, else the candidate-search arrays in the per-lane kernels (trees with octrees do not use
// the candidate search).
constexpr size_t lds_bytes(bool wave, int feat, int threads) {
    if (feat & FEAT_OCTREE) return (sizeof(uint2) * OCT_ANC_LEVELS + sizeof(uint32_t) * OCT_WALK_DWORDS) * (size_t)threads;
    if (!wave) return sizeof(uint32_t) * (CAND_WORDS + 1) * (size_t)threads + sizeof(uint16_t) * CAND_QUEUE * (size_t)(threads / 64);
    return 0;
}
// Waves per SIMD per kernel class; the macros exist so that a library can be built per setting and the
// settings compared on the same scenes (scripts/perf_variants.sh).
#ifndef MRT_LANE_WAVES
#define MRT_LANE_WAVES 3
#endif
#ifndef MRT_LEAN_WAVES
#define MRT_LEAN_WAVES 4
#endif
#ifndef MRT_HEAVY_WAVES
#define MRT_HEAVY_WAVES 3
#endif
#ifndef MRT_OCTREE_WAVES
#define MRT_OCTREE_WAVES 3
#endif
constexpr int seg_waves(bool wave, int feat) {
    // the per-lane kernels: two waves measured 9 % slower, a cap of 128 registers 31 % slower (spills), and their
    // LDS arrays allow three workgroups per CU
    if (!wave) return MRT_LANE_WAVES;
    // measured per feature set (with the loop's constants and arguments no longer hoisted into registers, and the
    // scene view read through a pointer): the wave-synchronous kernels without octree code gain from a fourth wave
    // (Cornell smoke +6 %, Easy Cornell Box +5 %, Light Demonstration +9 %) except the set with everything
    // (Perlin noise, media: Easy Scene -1.4 %); the octree walk loses a third of its speed to the spills of 128
    // registers
    if ((feat & FEAT_OCTREE) != 0) return MRT_OCTREE_WAVES;
    return (feat & FEAT_PERLIN) != 0 ? MRT_HEAVY_WAVES : MRT_LEAN_WAVES;
}

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool FROM_MEMORY>
__device__ __forceinline__ decltype(auto) loop_params(const RenderParams& by_value, const CONST_AS RenderParams* in_memory) {
    if constexpr (FROM_MEMORY) return (*in_memory);
    else return (by_value);
}
// One launch = up to `steps` segments for every slot of the pool.
template <bool WAVE, int FEAT>
__global__ void __launch_bounds__(seg_block(WAVE), seg_waves(WAVE, FEAT)) k_segments(const SceneView* svp, Pool pool, RenderParams rp,
                                                                               Counters* ctr, unsigned long long* wave_chunks) {
    SceneRef sv = *as_const(svp);
    // The render parameters are read where they are used, from the kernel-argument block itself (it is memory):
    // taken from the by-value argument they sit in scalar registers across the whole segment loop, get spilled
    // into vector-register lanes, and every use in the regeneration and accumulation code is a v_readlane.
    // `rpp` passes through an empty asm at the top of every loop turn so that the loads stay inside the loop.
    // (The wave-synchronous kernels only: Cornell smoke +0.7 %, Volume Ice +2 %, no SGPR spills left in the lean
    // kernels; the per-lane kernels measured 3 % slower with it and keep the argument.)
    const RenderParams& rp_by_value = rp;
    struct KernelArgs {
        const SceneView* svp;
        Pool pool;
        RenderParams rp;
        Counters* ctr;
        unsigned long long* wave_chunks;
    };
    const CONST_AS RenderParams* rpp = &((const CONST_AS KernelArgs*)__builtin_amdgcn_kernarg_segment_ptr())->rp;
    const uint32_t slot = blockIdx.x * seg_block(WAVE) + threadIdx.x;
    const bool in_range = slot < rp.n_slots;
    int32_t seg = -1;
    RayD ray{{0, 0, 0}, {0, 0, 1}, 0};
    F3 thr{0.f, 0.f, 0.f};
    uint32_t pixel = 0;
    uint64_t key = 0;
    if (in_range) {
        seg = pool.seg()[slot];
        if (seg >= 0) {
            ray.o = {pool.ox()[slot], pool.oy()[slot], pool.oz()[slot]};
            ray.d = {pool.dx()[slot], pool.dy()[slot], pool.dz()[slot]};
            ray.time = pool.tm()[slot];
            thr = {pool.tr()[slot], pool.tg()[slot], pool.tb()[slot]};
            pixel = pool.pixel()[slot];
            key = pool.key()[slot];
        }
    }
    uint32_t rays = 0, light_queries = 0;
    bool walking = false;  // this lane's octree walk was suspended in the previous turn of the loop (nearest_hit_wave)
    float next_cap = INFINITY;  // per-lane kernels: an upper bound on the answer of the path's next query (not kept across launches)
    uint32_t ended = 0;  // paths of this wave that ended in this launch (wave-uniform: a scalar register)
    // this wave's share of the work range (wave-uniform)
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(slot >> 6);
    unsigned long long chunk_next = wave_chunks[2 * (size_t)wave_id];
    unsigned long long chunk_end = wave_chunks[2 * (size_t)wave_id + 1];
    chunk_next = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(chunk_next >> 32)) << 32) |
                 __builtin_amdgcn_readfirstlane((uint32_t)chunk_next);
    chunk_end = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(chunk_end >> 32)) << 32) |
                __builtin_amdgcn_readfirstlane((uint32_t)chunk_end);
    bool dry = rp.all_claimed != 0;  // the global counter has run past the end of the work
    CandLds cand_bits{nullptr, nullptr, 0};
    // (kernels with octree code keep their LDS for the octree walk's ancestor cache)
    if constexpr (!WAVE && (FEAT & FEAT_OCTREE) == 0) cand_bits = candidate_lds(sv);
    PerlinLds perlin_lds{nullptr, false};
    if constexpr (WAVE && (FEAT & FEAT_PERLIN) != 0 && (FEAT & FEAT_OCTREE) == 0) {
        // (single-wave workgroups: the wave copies the table for itself, no barrier)
        __shared__ mrt_perlin perlin_table;
        static_assert(sizeof(mrt_perlin) % (4 * 64) == 0, "copied as whole rows of 64 dwords");
        if (sv.n_perlins > 0) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(sv.perlins);
            LDS_AS uint32_t* dst = (LDS_AS uint32_t*)reinterpret_cast<uint32_t*>(&perlin_table);
            for (uint32_t i = threadIdx.x; i < sizeof(mrt_perlin) / 4; i += seg_block(WAVE)) dst[i] = src[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            perlin_lds = {(const LDS_AS mrt_perlin*)&perlin_table, true};
        }
    }
    Stamps st{};
    REGION_BEGIN(total);
    const uint32_t steps = rp.steps;
    for (uint32_t step = 0; step < steps; ++step) {
        if constexpr (WAVE) asm volatile("" : "+s"(rpp));
        decltype(auto) rp = loop_params<WAVE>(rp_by_value, rpp);
        // ---- regeneration: free slots take the next pixel-samples of the wave's range
        REGION_BEGIN(regen);
        const bool want = in_range && seg < 0;
        const unsigned long long mask = __ballot(want);
        if (mask && !(dry && chunk_next >= chunk_end)) {
            const uint32_t n_want = (uint32_t)__popcll(mask);
            const uint32_t rank = (uint32_t)__popcll(mask & ((1ULL << lane_id()) - 1ULL));
            const unsigned long long avail = chunk_end - chunk_next;
            unsigned long long w = ~0ULL;
            if (avail >= n_want) {
                w = chunk_next + rank;
                chunk_next += n_want;
            } else {
                // serve what is left of the old range, then refill once (one atomic per wave)
                unsigned long long nb = rp.n_work, ne = rp.n_work;
                if (!dry) {
                    unsigned long long base = 0;
                    const int leader = __ffsll((long long)mask) - 1;
                    REGION_BEGIN(claim);
                    if ((int)lane_id() == leader) base = atomicAdd(&ctr->work_next, (unsigned long long)rp.chunk);
                    base = __shfl(base, leader, 64);
                    REGION_END(st, ST_CLAIM, claim);
                    if (base < rp.n_work) {
                        nb = base;
                        ne = base + rp.chunk < rp.n_work ? base + rp.chunk : rp.n_work;
                    } else {
                        dry = true;
                    }
                }
                if (rank < avail) {
                    w = chunk_next + rank;
                } else {
                    unsigned long long w2 = nb + (rank - avail);
                    if (w2 < ne) w = w2;
                }
                unsigned long long taken = n_want - avail;
                if (taken > ne - nb) taken = ne - nb;
                chunk_next = nb + taken;
                chunk_end = ne;
            }
            bool fresh = false;  // lanes that start a path in this turn
            if (want && w != ~0ULL) {
                uint32_t sample, pl, y, xs;
                divmod_u32(w, (uint32_t)rp.npix_strip, rp.inv_npix_strip, sample, pl);
                divmod_u32(pl, rp.strip_w, rp.inv_strip_w, y, xs);
                uint32_t x = rp.x0 + xs;
                pixel = x + y * rp.width;
                key = path_key(rp.seed, (uint64_t)pixel, rp.first_sample + sample);
                Rng g{key, 0};
                ray = camera_ray_pinhole(sv, x, y, rp.width, rp.height, g, rp.inv_w1, rp.inv_h1);
                thr = {1.f, 1.f, 1.f};
                seg = 0;
                fresh = true;
            }
            // The lens sample of the new paths, by the whole wave: a rejection loop per lane kept the wave for its
            // unluckiest lane (3.4 attempts of two draws with forty lanes asking; 10 % of the headline kernel).
            if (sv.cam.lens_radius != 0.0) {
                const D3 lens = wave_rand_reject<2>(fresh, key, MRT_DIM_LENS);
                if (fresh) camera_lens(sv, lens, ray);
            }
        }
        REGION_END(st, ST_REGEN, regen);
        const bool live = seg >= 0;
        if (WAVE && !__any(live)) break;
        Rng g{key, (uint32_t)(live ? seg : 0)};
        F3 radiance;
        // (an octree walk may be suspended when the lanes that are not walking can do something meanwhile: shade a
        // finished query, or start a new path; never in a launch's last turn, so that every launch finishes walks --
        // a walk still suspended when the launch ends starts again from the ray in the next one)
        const bool may_suspend = step + 1 < steps;
        bool cont = path_segment<WAVE, FEAT>(sv, ray, thr, g, live, radiance, rays, light_queries, st, cand_bits, perlin_lds, walking,
                                             may_suspend, next_cap);
        REGION_BEGIN(acc);
        const bool answered = live && !walking;  // (`walking`: the segment goes on in the next turn, nothing of the path moves)
        if (answered && cont && seg >= 49) cont = false;  // depth 50 -> 0 returns BLACK (ray_tracer.rs:166-172, :451)
        if (!(answered && cont)) next_cap = INFINITY;
        ended += (uint32_t)__popcll(__ballot(answered && !cont));
        if (answered) {
            seg++;
            if (!cont) {
                // ParallelImagePart::add_xy, parallel_image.rs:328-331
                float* px = rp.accum + (size_t)pixel * 3;
                if (radiance.r != 0.f) atomicAdd(px + 0, radiance.r);
                if (radiance.g != 0.f) atomicAdd(px + 1, radiance.g);
                if (radiance.b != 0.f) atomicAdd(px + 2, radiance.b);
                seg = -1;
            }
        }
        REGION_END(st, ST_ACCUM, acc);
        if (!__any(seg >= 0) && dry && chunk_next >= chunk_end) break;
    }
    REGION_END(st, ST_TOTAL, total);
#ifdef MRT_STAMPS
    for (int i = 0; i < ST_COUNT; ++i)
        if (st.acc[i]) atomicAdd(&ctr->stamps[i], st.acc[i]), atomicAdd(&ctr->stamp_lanes[i], st.lanes[i]);
    for (int i = 0; i < 12; ++i)
        if (st.count[i]) atomicAdd(&ctr->stamp_counts[i], (unsigned long long)st.count[i]);
#endif
    if (in_range) {
        pool.seg()[slot] = seg;
        if (seg >= 0) {
            pool.ox()[slot] = ray.o.x, pool.oy()[slot] = ray.o.y, pool.oz()[slot] = ray.o.z;
            pool.dx()[slot] = ray.d.x, pool.dy()[slot] = ray.d.y, pool.dz()[slot] = ray.d.z;
            pool.tm()[slot] = ray.time;
            pool.tr()[slot] = thr.r, pool.tg()[slot] = thr.g, pool.tb()[slot] = thr.b;
            pool.pixel()[slot] = pixel;
            pool.key()[slot] = key;
        }
    }
    uint32_t live = wave_sum(seg >= 0 ? 1u : 0u);
    rays = wave_sum(rays);
    light_queries = wave_sum(light_queries);
    if (lane_id() == 0) {
        wave_chunks[2 * (size_t)wave_id] = chunk_next;
        wave_chunks[2 * (size_t)wave_id + 1] = chunk_end;
        if (chunk_end > chunk_next) atomicAdd(&ctr->held, chunk_end - chunk_next);
        if (live) atomicAdd(&ctr->live, live);
        if (rays) atomicAdd(&ctr->rays, (unsigned long long)rays);
        if (light_queries) atomicAdd(&ctr->light_queries, (unsigned long long)light_queries);
        if (ended) atomicAdd(&ctr->paths, (unsigned long long)ended);
    }
}

// ---- the two single-bounce shaders (CurrentShader::Diffuse / LightMap, ray_tracer.rs:304-309)
// DiffuseShader::ray_color (ray_tracer.rs:126-157): what HitRecord::new (hittable.rs:414-448) decides
// for the nearest hit -- emitted colour, scatter attenuation, or black -- or the background.
template <int FEAT>
__device__ F3 diffuse_stage(SceneRef sv, const RayD& ray, const HitD& h, bool hit) {
    if (!hit) return background_color<FEAT>(sv, ray);
    const DObjRec* ob = sv.objects + h.obj;
    Surface s;
    surface_at<FEAT>(ob, ray, h.t, h.sub, h.pos, s);
    const F3 black{0.f, 0.f, 0.f};
    if constexpr ((FEAT & FEAT_OCTREE) != 0) {
        if (s.voxel >= 0) {  // VoxelMaterial::scatter, voxel.rs:83-117
            F3 att;
            double fuzz;
            int mk = voxel_material<FEAT>(sv, s, att, fuzz);
            if (mk == MRT_MAT_METAL) {
                D3 reflected = reflect(normalize(s.dir_obj), s.n_obj);
                return dot(reflected, s.n_obj) > 0.0 ? att : black;
            }
            return att;
        }
    }
    switch (ob->mat_kind) {
        case MRT_MAT_DIFFUSE_LIGHT: return s.front_face ? texture_color<FEAT>(sv, ob, s) : black;  // material.rs:236-248
        case MRT_MAT_METAL: {  // material.rs:105-124: no scatter record when the reflection points inward
            D3 reflected = reflect(normalize(s.dir_obj), s.n_obj);
            return dot(reflected, s.n_obj) > 0.0 ? texture_color<FEAT>(sv, ob, s) : black;
        }
        case MRT_MAT_DIELECTRIC: return {ob->color[0], ob->color[1], ob->color[2]};  // material.rs:190-195
        default: return texture_color<FEAT>(sv, ob, s);  // Lambertian, Isotropic
    }
}

// One thread per pixel-sample, no path pool: camera ray, nearest hit, and for the light map one
// sampled point and one more nearest-hit query per light (LightMapShader::ray_color,
// ray_tracer.rs:83-124; segment 1 + i addresses the random numbers of light i).
template <bool WAVE, int FEAT, int MODE>
__global__ void __launch_bounds__(BLOCK) k_single_bounce(const SceneView* svp, RenderParams rp, unsigned long long w_begin,
                                                         unsigned long long w_end, Counters* ctr) {
    SceneRef sv = *as_const(svp);
    const unsigned long long w = w_begin + (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = w < w_end;
    RayD ray{{0, 0, 0}, {0, 0, 1}, 0};
    uint32_t pixel = 0;
    uint64_t key = 0;
    if (live) {
        uint32_t sample, pl, y, xs;
        divmod_u32(w, (uint32_t)rp.npix_strip, rp.inv_npix_strip, sample, pl);
        divmod_u32(pl, rp.strip_w, rp.inv_strip_w, y, xs);
        uint32_t x = rp.x0 + xs;
        pixel = x + y * rp.width;
        key = path_key(rp.seed, (uint64_t)pixel, rp.first_sample + sample);
        ray = camera_ray(sv, x, y, rp.width, rp.height, Rng(key, 0), rp.inv_w1, rp.inv_h1);
    }
    Stamps st{};
    uint32_t rays = live ? 1u : 0u;
    HitD h{0.0, -1, 0, {0.0, 0.0, 0.0}};
    Rng g0(key, 0);
    CandLds cand_bits{nullptr, nullptr, 0};
    if constexpr (!WAVE && (FEAT & FEAT_OCTREE) == 0) cand_bits = candidate_lds(sv);
    bool hit = WAVE ? nearest_hit_wave<FEAT>(sv, ray, g0, live, h, st) : nearest_hit_lane<FEAT>(sv, ray, g0, live, h, st, cand_bits);
    F3 color{0.f, 0.f, 0.f};
    if constexpr (MODE == MRT_SHADER_DIFFUSE) {
        if (live) color = diffuse_stage<FEAT>(sv, ray, h, hit);
    } else {
        constexpr int LF = FEAT & ~(FEAT_OCTREE | FEAT_VOLUME);
        hit = hit && live;
        D3 pos_w{0, 0, 0};
        if (hit) {
            const DObjRec* ob = sv.objects + h.obj;
            Surface s;
            surface_at<FEAT>(ob, ray, h.t, h.sub, h.pos, s);
            pos_w = point_to_world(ob, s.pos_obj);
        }
        auto lights = as_const(sv.lights);
        for (int i = 0; i < sv.n_lights; ++i) {
            Rng gi(key, 1u + (uint32_t)i);
            AreaInfo ai;
            ai.to_area = ray;
            ai.end_point = {0, 0, 0};
            if (hit) object_generate<LF>(lights + i, pos_w, h.t, gi, rng_f64(gi, MRT_DIM_LIGHT_A), rng_f64(gi, MRT_DIM_LIGHT_B), ai);
            HitD h2{0.0, -1, 0, {0.0, 0.0, 0.0}};
            bool hit2 = WAVE ? nearest_hit_wave<FEAT>(sv, ai.to_area, gi, hit, h2, st)
                             : nearest_hit_lane<FEAT>(sv, ai.to_area, gi, hit, h2, st, cand_bits);
            if (hit) rays++;
            if (hit && hit2) {
                const DObjRec* ob2 = sv.objects + h2.obj;
                Surface s2;
                surface_at<FEAT>(ob2, ai.to_area, h2.t, h2.sub, h2.pos, s2);
                D3 t = ai.end_point - point_to_world(ob2, s2.pos_obj);
                float o = (float)magnitude(t);  // m * RgbColor::WHITE narrows m to f32 (rgb_color.rs:138-147)
                if (o != o) color.r += 1.f;     // is_nan -> RgbColor::new(1, 0, 0)
                else color.r += o, color.g += o, color.b += o;
            }
        }
    }
    if (live) {
        float* px = rp.accum + (size_t)pixel * 3;
        if (color.r != 0.f) atomicAdd(px + 0, color.r);
        if (color.g != 0.f) atomicAdd(px + 1, color.g);
        if (color.b != 0.f) atomicAdd(px + 2, color.b);
    }
    const uint32_t ended = (uint32_t)__popcll(__ballot(live));
    rays = wave_sum(rays);
    if (lane_id() == 0 && rays) atomicAdd(&ctr->rays, (unsigned long long)rays);
    if (lane_id() == 0 && ended) atomicAdd(&ctr->paths, (unsigned long long)ended);
}

__global__ void __launch_bounds__(BLOCK) k_init_pool(Pool pool, uint32_t n, unsigned long long* wave_chunks) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) pool.seg()[i] = -1;
    if (i < n && (i & 63) == 0) wave_chunks[2 * (size_t)(i >> 6)] = 0, wave_chunks[2 * (size_t)(i >> 6) + 1] = 0;
}

// Stream compaction: live slots of `src` are packed to the front of `dst`
// (wave64 ballot + popcount prefix, one atomic per wave).
__global__ void __launch_bounds__(BLOCK) k_compact(Pool src, Pool dst, uint32_t n, Counters* ctr) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    bool alive = i < n && src.seg()[i] >= 0;
    unsigned long long mask = __ballot(alive);
    if (!mask) return;
    uint32_t base = 0;
    int leader = __ffsll((long long)mask) - 1;
    if ((int)lane_id() == leader) base = atomicAdd(&ctr->compacted, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (alive) {
        uint32_t j = base + (uint32_t)__popcll(mask & ((1ULL << lane_id()) - 1ULL));
        dst.ox()[j] = src.ox()[i], dst.oy()[j] = src.oy()[i], dst.oz()[j] = src.oz()[i];
        dst.dx()[j] = src.dx()[i], dst.dy()[j] = src.dy()[i], dst.dz()[j] = src.dz()[i];
        dst.tm()[j] = src.tm()[i];
        dst.tr()[j] = src.tr()[i], dst.tg()[j] = src.tg()[i], dst.tb()[j] = src.tb()[i];
        dst.pixel()[j] = src.pixel()[i];
        dst.key()[j] = src.key()[i];
        dst.seg()[j] = src.seg()[i];
    }
}

// Probe kernels (parity tests).
__global__ void __launch_bounds__(BLOCK) k_camera_rays(const SceneView* svp, uint32_t width, uint32_t height,
                                                       const int32_t* pixels, uint32_t n, uint64_t seed,
                                                       uint64_t sample, double* out) {
    SceneRef sv = *as_const(svp);
    // (the segment kernel's way: the pinhole ray per lane, the lens sample by the whole wave -- odd lanes; the even
    // lanes take the one-lane form the single-bounce shaders use, so that the probe checks both)
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool live = i < n;
    uint32_t x = 0, y = 0;
    if (live) x = (uint32_t)pixels[2 * i], y = (uint32_t)pixels[2 * i + 1];
    const uint64_t key = path_key(seed, (uint64_t)x + (uint64_t)y * width, sample);
    Rng g{key, 0};
    const double inv_w1 = reciprocal_for_div_by((double)width - 1.0), inv_h1 = reciprocal_for_div_by((double)height - 1.0);
    const bool by_wave = (i & 1u) != 0;
    RayD r = camera_ray_pinhole(sv, x, y, width, height, g, inv_w1, inv_h1);
    if (sv.cam.lens_radius != 0.0) {
        const D3 lens = wave_rand_reject<2>(live && by_wave, key, MRT_DIM_LENS);
        if (live && by_wave) camera_lens(sv, lens, r);
    }
    if (live && !by_wave) r = camera_ray(sv, x, y, width, height, g, inv_w1, inv_h1);
    if (!live) return;
    double* o = out + 7 * (size_t)i;
    o[0] = r.o.x, o[1] = r.o.y, o[2] = r.o.z, o[3] = r.d.x, o[4] = r.d.y, o[5] = r.d.z, o[6] = r.time;
}
template <bool WAVE, int FEAT>
__global__ void __launch_bounds__(BLOCK) k_trace_rays(const SceneView* svp, const double* rays, uint32_t n, uint64_t seed,
                                                      int32_t* obj_id, double* t_out, double* normal,
                                                      double* position, int32_t* front_face) {
    SceneRef sv = *as_const(svp);
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool live = i < n;
    RayD ray{{0, 0, 0}, {0, 0, 1}, 0};
    if (live) {
        const double* r = rays + 7 * (size_t)i;
        ray = {{r[0], r[1], r[2]}, {r[3], r[4], r[5]}, r[6]};
    }
    Rng g{path_key(seed, (uint64_t)i, 0), 0};
    HitD h{0.0, -1, 0, {0.0, 0.0, 0.0}};
    Stamps st{};
    CandLds cand_bits{nullptr, nullptr, 0};
    // (kernels with octree code keep their LDS for the octree walk's ancestor cache)
    if constexpr (!WAVE && (FEAT & FEAT_OCTREE) == 0) cand_bits = candidate_lds(sv);
    bool hit = WAVE ? nearest_hit_wave<FEAT>(sv, ray, g, live, h, st) : nearest_hit_lane<FEAT>(sv, ray, g, live, h, st, cand_bits);
    if (!live) return;
    D3 n_w{0, 0, 0}, p_w{0, 0, 0};
    if (hit) {
        const DObjRec* ob = sv.objects + h.obj;
        Surface s;
        surface_at<FEAT>(ob, ray, h.t, h.sub, h.pos, s);
        n_w = normal_to_world(ob, s.n_obj);
        p_w = point_to_world(ob, s.pos_obj);
    }
    if (obj_id) obj_id[i] = hit ? h.obj : -1;
    if (t_out) t_out[i] = hit ? h.t : 0.0;
    if (normal) normal[3 * (size_t)i] = n_w.x, normal[3 * (size_t)i + 1] = n_w.y, normal[3 * (size_t)i + 2] = n_w.z;
    if (position) position[3 * (size_t)i] = p_w.x, position[3 * (size_t)i + 1] = p_w.y, position[3 * (size_t)i + 2] = p_w.z;
    if (front_face) front_face[i] = hit ? (dot(ray.d, n_w) <= 0.0 ? 1 : 0) : 0;  // hittable.rs:222
}

}  // namespace mrt

// ====================================================================== host side
using namespace mrt;

// ---- device memory of mrt_render_accumulate_device_noalloc, the render entry point that allocates
// nothing: one path pool of NOALLOC_SLOTS slots (8192 waves = two generations of the waves a 256-CU
// GPU holds at four per SIMD; with exactly one generation about an eighth of the wave slots stays
// empty, measured), the counters and the per-wave work ranges, in a file-scope device array that
// every GPU the library runs on holds until the process ends.
// In host code the array's name is only a handle for hipGetSymbolAddress; the address the kernels
// get is the per-device pointer resolved by resolve_scratch().
constexpr uint32_t NOALLOC_SLOTS = 8192u * 64u;
constexpr size_t align256(size_t n) { return (n + 255) & ~size_t(255); }
struct ScratchLayout {
    size_t pool, counters, wave_chunks, total;
};
constexpr ScratchLayout scratch_layout() {
    ScratchLayout l{};
    size_t off = 0;
    l.pool = off, off += align256(Pool::bytes(NOALLOC_SLOTS));
    l.counters = off, off += align256(sizeof(Counters));
    l.wave_chunks = off, off += align256(sizeof(unsigned long long) * 2 * (NOALLOC_SLOTS / 64));
    l.total = off;
    return l;
}
constexpr ScratchLayout SCRATCH = scratch_layout();
static_assert(NOALLOC_SLOTS == MRT_NOALLOC_POOL_SLOTS && SCRATCH.total == MRT_NOALLOC_SCRATCH_BYTES,
              "include/mrt.h states the pool slots and the bytes of the scratch array");
static __device__ __align__(256) unsigned char scratch[SCRATCH.total];

namespace {

thread_local std::string g_last_error;

int set_error(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
}  // namespace
void mrt::set_last_error(const std::string& msg) { g_last_error = msg; }
namespace {
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return set_error(e_ == hipErrorOutOfMemory ? MRT_ERR_OUT_OF_MEMORY : MRT_ERR_DEVICE,       \
                             std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

struct DeviceBuffers {
    std::vector<void*> allocs;
    ~DeviceBuffers() {
        for (void* p : allocs) (void)hipFree(p);
    }
    template <class T>
    int upload(const std::vector<T>& v, const T** out) {
        *out = nullptr;
        size_t bytes = v.size() * sizeof(T);
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
        allocs.push_back(p);
        if (bytes) HIP_TRY(hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice));
        *out = static_cast<const T*>(p);
        return MRT_OK;
    }
    int alloc(size_t bytes, void** out) {
        *out = nullptr;
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
        allocs.push_back(p);
        *out = p;
        return MRT_OK;
    }
};

}  // namespace

struct mrt_scene {
    FlatScene flat;
    int device = -1;  // -1: host-only scene (no device image)
    SceneView view{};
    DeviceBuffers scene_mem;
    // path pool (allocated lazily on the first render)
    std::unique_ptr<DeviceBuffers> pool_mem;
    Pool pool[2]{};
    Counters* counters = nullptr;
    unsigned long long* wave_chunks = nullptr;  // [next, end) of the work range each wave holds
    uint32_t pool_slots = 0;
    // options
    int64_t opt_pool_slots = 1 << 21;
    int64_t opt_steps = 4096;
    int64_t opt_trace_zero = 0;
    int64_t opt_compact = 1;
    int64_t opt_shader = MRT_SHADER_RAYTRACING;
    double paths_per_second = 0.0;  // measured on the scene's latest launch that handed out work
    int64_t opt_oct_suspend = 1;  // 0: octree walks are never suspended (same results; for comparison)
    int64_t opt_traversal = 0;  // 0 auto, 1 per-lane, 2 wave-synchronous over the tree, 3 over the leaf list, 4 candidates
    const DNodeRec* dev_node_recs = nullptr;
    const DNodeRec* dev_leaf_recs = nullptr;
    // what the kernels get: device copies of the view, one per traversal mode and setting of
    // "trace_zero_throughput" (device_view), written when the scene is created
    SceneView* dev_views = nullptr;
    int features = 0;           // FEAT_* bits the scene needs
    // per-scene call resources, created with the scene so that a render call creates nothing:
    // events (whole call begin / end, segment kernel begin / end) and a pinned host copy of the counters
    hipEvent_t events[4]{};
    int n_events = 0;
    Counters* host_counters = nullptr;
    // accumulation image of mrt_render_accumulate_multi (resident between calls)
    float* multi_accum = nullptr;
    size_t multi_accum_floats = 0;
    std::mutex mu;
    ~mrt_scene() {
        for (int i = 0; i < n_events; ++i) (void)hipEventDestroy(events[i]);
        if (host_counters) (void)hipHostFree(host_counters);
        if (multi_accum) (void)hipFree(multi_accum);
    }
};

// Feature sets the kernels are instantiated for, smallest first; the last one covers everything.
template <int... M>
struct FeatList {};
constexpr int FEAT_NO_OCTREE = FEAT_ALL & ~(FEAT_OCTREE | FEAT_VOLUME);
using KernelFeats = FeatList<0, FEAT_MEDIUM, FEAT_SPHERE | FEAT_MEDIUM, FEAT_SPHERE | FEAT_SPHERE_LEAF | FEAT_MOVING | FEAT_SPECULAR | FEAT_SKY,
                             FEAT_SPHERE | FEAT_SPHERE_LEAF | FEAT_SPECULAR | FEAT_TEXTURE | FEAT_SKY,
                             FEAT_NO_OCTREE, FEAT_OCTREE | FEAT_SKY, FEAT_OCTREE | FEAT_SPHERE | FEAT_SPHERE_LEAF | FEAT_SKY,
                             FEAT_OCTREE | FEAT_VOLUME | FEAT_SPHERE | FEAT_SPHERE_LEAF | FEAT_SKY, FEAT_ALL>;
// The single-bounce shaders (debug views) are instantiated for two sets only.
using SingleBounceFeats = FeatList<FEAT_NO_OCTREE, FEAT_ALL>;
// Calls f(std::integral_constant<int, M>) for the first instantiated set M that covers `need`.
template <class F, int M0, int... M>
static void dispatch_feat(int need, FeatList<M0, M...>, F&& f) {
    if ((need & ~M0) == 0 || sizeof...(M) == 0) {
        f(std::integral_constant<int, M0>{});
        return;
    }
    if constexpr (sizeof...(M) > 0) dispatch_feat(need, FeatList<M...>{}, f);
}
static int scene_features(const FlatScene& f) {
    int need = 0;
    for (const DShape& sh : f.shapes) {
        if (sh.kind == MRT_SHAPE_OCTREE) need |= FEAT_OCTREE;
        if (sh.kind == MRT_SHAPE_SPHERE || sh.kind == MRT_SHAPE_MOVING_SPHERE) need |= FEAT_SPHERE;
        if (sh.kind == MRT_SHAPE_MOVING_SPHERE) need |= FEAT_MOVING;
        if (sh.kind == MRT_SHAPE_CONSTANT_MEDIUM) need |= FEAT_MEDIUM;
    }
    for (const DObject& ob : f.objects)
        if (f.shapes[(size_t)ob.shape].kind == MRT_SHAPE_SPHERE && ob.identity) need |= FEAT_SPHERE_LEAF;
    for (const DMaterial& m : f.materials)
        if (m.kind == MRT_MAT_METAL || m.kind == MRT_MAT_DIELECTRIC) need |= FEAT_SPECULAR;
    for (const DTexture& t : f.textures)
        if (t.kind != MRT_TEX_SOLID) need |= FEAT_TEXTURE | (t.kind == MRT_TEX_PERLIN ? FEAT_PERLIN : 0);
    if (f.background.kind != MRT_BG_CONSTANT || f.sun.has_sun) need |= FEAT_SKY;
    for (const mrt_voxel& v : f.oct_voxels)
        if (v.kind != MRT_VOX_LAMBERTIAN) need |= FEAT_VOLUME;
    return need;
}

// Trees of at most WAVE_SYNC_MAX_NODES nodes are walked wave-synchronously; with at most
// LEAF_LIST_MAX leaves the walk visits the leaf list instead of the tree.
constexpr int WAVE_SYNC_MAX_NODES = 64;
constexpr int LEAF_LIST_MAX = 16;
enum : int { TRAV_LANE = 1, TRAV_WAVE_TREE = 2, TRAV_WAVE_LEAVES = 3, TRAV_CANDIDATES = 4 };
static int traversal_mode(const mrt_scene* sc) {
    const bool have_list = !sc->flat.leaf_recs.empty();
    const bool have_cand = !sc->flat.cand_nodes.empty() && (sc->features & FEAT_OCTREE) == 0;
    if (sc->opt_traversal == TRAV_LANE || sc->opt_traversal == TRAV_WAVE_TREE) return (int)sc->opt_traversal;
    if (sc->opt_traversal == TRAV_WAVE_LEAVES) return have_list ? TRAV_WAVE_LEAVES : TRAV_WAVE_TREE;
    if (sc->opt_traversal == TRAV_CANDIDATES) return have_cand ? TRAV_CANDIDATES : TRAV_LANE;
    if ((int)sc->flat.node_object.size() > WAVE_SYNC_MAX_NODES) return have_cand ? TRAV_CANDIDATES : TRAV_LANE;
    return have_list && (int)sc->flat.leaf_recs.size() <= LEAF_LIST_MAX ? TRAV_WAVE_LEAVES : TRAV_WAVE_TREE;
}
// Scene view for one launch: points the wave-synchronous walk at the tree or at the leaf list.
static SceneView view_for(const mrt_scene* sc, int mode) {
    SceneView v = sc->view;
    if (mode != TRAV_CANDIDATES) v.n_cand_nodes = 0, v.n_cand_words = 0;
    const std::vector<DNodeRec>* recs = &sc->flat.node_recs;
    if (mode == TRAV_WAVE_LEAVES) {
        v.node_recs = sc->dev_leaf_recs;
        v.n_wave_nodes = (int32_t)sc->flat.leaf_recs.size();
        recs = &sc->flat.leaf_recs;
    } else {
        v.node_recs = sc->dev_node_recs;
        v.n_wave_nodes = (int32_t)sc->flat.node_recs.size();
    }
    // the first node of the wave-synchronous walk is a leaf with a solid-voxel octree that keeps a remembered path:
    // its walk can be suspended (octree_first_leaf). Scenes with volume voxels run kernels without that code.
    v.oct_first = 0;
    if ((mode == TRAV_WAVE_LEAVES || mode == TRAV_WAVE_TREE) && !recs->empty() && (sc->features & FEAT_VOLUME) == 0 &&
        sc->opt_oct_suspend != 0) {
        const DNodeRec& r0 = (*recs)[0];
        if (r0.object >= 0 && r0.kind == MRT_SHAPE_OCTREE && (r0.flags & (NODE_TWICE | NODE_DYNAMIC_BOX)) == 0 && r0.octree >= 0 &&
            (size_t)r0.octree < sc->flat.octrees.size() && sc->flat.octrees[(size_t)r0.octree].cached != 0)
            v.oct_first = 1;
    }
    return v;
}

namespace {

// Writes the device copies of the scene view (the scene's device is current).
int upload_views(mrt_scene* sc) {
    for (int mode = TRAV_LANE; mode <= TRAV_CANDIDATES; ++mode) {
        for (int tz = 0; tz < 2; ++tz) {
            SceneView dv = view_for(sc, mode);
            dv.trace_zero_throughput = tz;
            HIP_TRY(hipMemcpy(sc->dev_views + (mode - 1) * 2 + tz, &dv, sizeof(SceneView), hipMemcpyHostToDevice));
        }
    }
    return MRT_OK;
}

int upload_scene(mrt_scene* sc) {
    FlatScene& f = sc->flat;
    SceneView& v = sc->view;
    DeviceBuffers& m = sc->scene_mem;
    int rc;
#define UP(vec, field)                                       \
    if ((rc = m.upload(vec, &v.field)) != MRT_OK) return rc;
    UP(f.nodes, nodes) UP(f.lane_leaf, lane_leaf)
    if ((rc = m.upload(f.node_recs, &sc->dev_node_recs)) != MRT_OK) return rc;
    if ((rc = m.upload(f.leaf_recs, &sc->dev_leaf_recs)) != MRT_OK) return rc;
    UP(f.cand_nodes, cand_nodes)
    UP(f.obj_recs, objects) UP(f.light_recs, lights)
    UP(f.textures, textures) UP(f.images, images) UP(f.image_pool, image_pool) UP(f.perlins, perlins)
    UP(f.octrees, octrees) UP(f.oct_nodes, oct_nodes) UP(f.oct_voxels, oct_voxels)
#undef UP
    v.n_nodes = (int32_t)f.node_object.size();
    v.n_cand_nodes = (int32_t)f.cand_nodes.size();
    v.n_cand_words = f.cand_nodes.empty() ? 0 : (int32_t)((f.node_object.size() + 31) / 32);
    v.n_objects = (int32_t)f.objects.size();
    v.n_lights = (int32_t)f.lights.size();
    v.n_perlins = (int32_t)f.perlins.size();
    v.cam = f.camera;
    v.bg = f.background;
    v.sun = f.sun;
    {
        const double r = std::cos(v.bg.sun_phi);
        v.bg_sun_ray[0] = r * std::cos(v.bg.sun_theta), v.bg_sun_ray[1] = std::sin(v.bg.sun_phi), v.bg_sun_ray[2] = r * std::sin(v.bg.sun_theta);
        v.bg_cos_sun_radius = std::cos(v.bg.sun_radius);
        const double rs = std::cos(v.sun.phi);
        v.sun_dir[0] = rs * std::cos(v.sun.theta), v.sun_dir[1] = std::sin(v.sun.phi), v.sun_dir[2] = rs * std::sin(v.sun.theta);
        v.sun_cos_radius = std::cos(v.sun.radius);
    }
    v.trace_zero_throughput = 0;
    v.cand_queue = CAND_QUEUE;
    if ((rc = m.alloc(sizeof(SceneView) * 2 * TRAV_CANDIDATES, reinterpret_cast<void**>(&sc->dev_views))) != MRT_OK) return rc;
    return upload_views(sc);
}
// The device copy of view_for(sc, mode) with "trace_zero_throughput" set as given.
const SceneView* device_view(const mrt_scene* sc, int mode, bool trace_zero) {
    return sc->dev_views + (mode - 1) * 2 + (trace_zero ? 1 : 0);
}

int ensure_pool(mrt_scene* sc) {
    uint32_t want = (uint32_t)std::max<int64_t>(BLOCK, std::min<int64_t>(sc->opt_pool_slots, 1 << 26));
    want = (want + BLOCK - 1) / BLOCK * BLOCK;
    if (sc->pool_mem && sc->pool_slots == want) return MRT_OK;
    sc->pool_mem.reset(new DeviceBuffers());
    DeviceBuffers& m = *sc->pool_mem;
    int rc;
    for (int b = 0; b < 2; ++b) {
        Pool& p = sc->pool[b];
        p.cap = want;
        if ((rc = m.alloc(Pool::bytes(want), reinterpret_cast<void**>(&p.base))) != MRT_OK) return rc;
    }
    if ((rc = m.alloc(sizeof(Counters), reinterpret_cast<void**>(&sc->counters))) != MRT_OK) return rc;
    if ((rc = m.alloc(sizeof(unsigned long long) * 2 * (want / 64), reinterpret_cast<void**>(&sc->wave_chunks))) != MRT_OK) return rc;
    sc->pool_slots = want;
    return MRT_OK;
}

// Per-device address of `scratch` (each GPU holds its own copy of the array at its own address),
// resolved once per device with that device current, and a lock per device: every call on a GPU
// shares that GPU's array, so calls of the no-allocation entry point on one GPU take turns.
constexpr int MAX_DEVICES = 64;
struct ScratchDevice {
    unsigned char* ptr = nullptr;
    hipError_t status = hipSuccess;
    bool resolved = false;
    std::mutex in_use;
};
ScratchDevice g_scratch_dev[MAX_DEVICES];
std::mutex g_scratch_mu;
// Requires `device` to be the current device.
hipError_t resolve_scratch(int device) {
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    ScratchDevice& d = g_scratch_dev[device];
    if (!d.resolved) {
        void* p = nullptr;
        d.status = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        d.ptr = d.status == hipSuccess ? static_cast<unsigned char*>(p) : nullptr;
        d.resolved = true;
    }
    return d.status;
}

// Events and the pinned counter copy of a scene (created once, with the scene's device current).
int create_call_resources(mrt_scene* sc) {
    for (; sc->n_events < 4; ++sc->n_events) HIP_TRY(hipEventCreate(&sc->events[sc->n_events]));
    if (!sc->host_counters) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sc->host_counters), sizeof(Counters), hipHostMallocDefault));
    return MRT_OK;
}

// Render loop on `stream`; accum is device memory. With `scratch_ptr` (this device's address of
// `scratch`) the pool, the counters and the work ranges are carved from it and nothing is allocated;
// otherwise they are the scene's own (allocated on its first render).
int render_device(mrt_scene* sc, float* accum_dev, uint32_t width, uint32_t height, uint32_t x0, uint32_t strip_w,
                  uint32_t spp, uint64_t seed, uint64_t first_sample, hipStream_t stream, mrt_stats* stats,
                  unsigned char* scratch_ptr = nullptr) {
    if (!sc || sc->device < 0) return set_error(MRT_ERR_INVALID_ARGUMENT, "scene has no device image");
    if (!accum_dev) return set_error(MRT_ERR_INVALID_ARGUMENT, "accumulation buffer is NULL");
    if (width < 2 || height < 2) return set_error(MRT_ERR_INVALID_ARGUMENT, "image must be at least 2x2");
    if ((uint64_t)width * height > 0xFFFFFFFFull / 3) return set_error(MRT_ERR_INVALID_ARGUMENT, "image too large");
    if (strip_w == 0 || (uint64_t)x0 + strip_w > width)
        return set_error(MRT_ERR_INVALID_ARGUMENT, "strip exceeds the image width");
    if ((unsigned long long)strip_w * height * (unsigned long long)spp >= (1ull << 53))
        return set_error(MRT_ERR_INVALID_ARGUMENT, "more than 2^53 paths in one call");
    std::lock_guard<std::mutex> lock(sc->mu);
    HIP_TRY(hipSetDevice(sc->device));
    {
        // The accumulation adds are hardware f32 atomics, which are only defined on ordinary
        // (coarse-grained) device memory of the scene's GPU: host, managed or another GPU's memory
        // would silently lose samples.
        hipPointerAttribute_t attr{};
        const hipError_t e = hipPointerGetAttributes(&attr, accum_dev);
        if (e != hipSuccess) (void)hipGetLastError();
        if (e != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != sc->device)
            return set_error(MRT_ERR_INVALID_ARGUMENT,
                             "the accumulation buffer must be hipMalloc'ed device memory on the scene's device");
    }
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (spp == 0) return MRT_OK;
    Pool pools[2]{};
    Counters* counters = nullptr;
    unsigned long long* wave_chunks = nullptr;
    uint32_t pool_slots = 0;
    if (scratch_ptr) {
        pools[0].base = scratch_ptr + SCRATCH.pool;
        pools[0].cap = NOALLOC_SLOTS;
        counters = reinterpret_cast<Counters*>(scratch_ptr + SCRATCH.counters);
        wave_chunks = reinterpret_cast<unsigned long long*>(scratch_ptr + SCRATCH.wave_chunks);
        pool_slots = NOALLOC_SLOTS;
    } else {
        int rc = ensure_pool(sc);
        if (rc != MRT_OK) return rc;
        pools[0] = sc->pool[0], pools[1] = sc->pool[1];
        counters = sc->counters, wave_chunks = sc->wave_chunks, pool_slots = sc->pool_slots;
    }
    const bool can_compact = !scratch_ptr && sc->opt_compact;  // the scratch holds one pool: no ping-pong partner
    if (sc->n_events < 4 || !sc->host_counters) return set_error(MRT_ERR_DEVICE, "scene has no call resources");
    const hipEvent_t ev_begin = sc->events[0], ev_end = sc->events[1], ek0 = sc->events[2], ek1 = sc->events[3];
    Counters& host = *sc->host_counters;

    RenderParams rp{};
    rp.accum = accum_dev;
    rp.width = width, rp.height = height, rp.x0 = x0, rp.strip_w = strip_w;
    rp.npix_strip = (unsigned long long)strip_w * height;
    rp.n_work = rp.npix_strip * spp;
    rp.inv_npix_strip = 1.0 / (double)rp.npix_strip;
    rp.inv_strip_w = 1.0 / (double)strip_w;
    rp.inv_w1 = reciprocal_for_div_by((double)width - 1.0), rp.inv_h1 = reciprocal_for_div_by((double)height - 1.0);
    rp.seed = seed, rp.first_sample = first_sample;
    rp.steps = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(sc->opt_steps, 65536));
    uint32_t n_slots = (uint32_t)std::min<unsigned long long>(pool_slots, (rp.n_work + BLOCK - 1) / BLOCK * BLOCK);
    // Claim size. Two pulls: small claims keep the waves finishing together (only a wave's own lanes
    // start the items it holds, so with 512-item claims a small render is in effect partitioned
    // statically and waits for its slowest waves), but every claim is an atomic on ONE counter, which
    // the memory system serves at some tens of millions per second. So the claim is what keeps the
    // claim rate near 8 M/s at the throughput the scene has shown on its latest launch, between one
    // wave's worth and WORK_CHUNK. Before a scene's first launch: an eighth of a wave's share of the
    // render (corrected after that launch).
    auto claim_size = [&](double paths_per_second) {
        unsigned long long c = (unsigned long long)(paths_per_second / 8.0e6);
        c = (c + 63ull) & ~63ull;
        return (uint32_t)(c < 64 ? 64 : (c > WORK_CHUNK ? WORK_CHUNK : c));
    };
    if (sc->paths_per_second > 0.0) {
        rp.chunk = claim_size(sc->paths_per_second);
    } else {
        const unsigned long long per_wave = rp.n_work / (8ull * (n_slots / 64u) + 1ull);
        rp.chunk = per_wave >= WORK_CHUNK ? WORK_CHUNK : (per_wave <= 64 ? 64u : (uint32_t)(per_wave & ~63ull));
    }
    const int trav = traversal_mode(sc);
    const SceneView* view = device_view(sc, trav, sc->opt_trace_zero != 0);
    const bool wave_sync = trav != TRAV_LANE && trav != TRAV_CANDIDATES;

    HIP_TRY(hipEventRecord(ev_begin, stream));
    HIP_TRY(hipMemsetAsync(counters, 0, sizeof(Counters), stream));
    if (sc->opt_shader != MRT_SHADER_RAYTRACING) {
        // single-bounce shaders: one thread per pixel-sample, in launches of at most 2^28 threads
        const unsigned long long per_launch = 1ull << 28;
        uint64_t n_launch = 0;
        HIP_TRY(hipEventRecord(ek0, stream));
        for (unsigned long long w0 = 0; w0 < rp.n_work; w0 += per_launch, ++n_launch) {
            const unsigned long long w1 = std::min(rp.n_work, w0 + per_launch);
            const dim3 grid((unsigned)((w1 - w0 + BLOCK - 1) / BLOCK)), block(BLOCK);
            const bool light_map = sc->opt_shader == MRT_SHADER_LIGHT_MAP;
            dispatch_feat(sc->features, SingleBounceFeats{}, [&](auto feat) {
                constexpr int FT = decltype(feat)::value;
                if (wave_sync && light_map)
                    hipLaunchKernelGGL((k_single_bounce<true, FT, MRT_SHADER_LIGHT_MAP>), grid, block, lds_bytes(true, FT, BLOCK), stream, view, rp, w0, w1, counters);
                else if (wave_sync)
                    hipLaunchKernelGGL((k_single_bounce<true, FT, MRT_SHADER_DIFFUSE>), grid, block, lds_bytes(true, FT, BLOCK), stream, view, rp, w0, w1, counters);
                else if (light_map)
                    hipLaunchKernelGGL((k_single_bounce<false, FT, MRT_SHADER_LIGHT_MAP>), grid, block, lds_bytes(false, FT, BLOCK), stream, view, rp, w0, w1, counters);
                else
                    hipLaunchKernelGGL((k_single_bounce<false, FT, MRT_SHADER_DIFFUSE>), grid, block, lds_bytes(false, FT, BLOCK), stream, view, rp, w0, w1, counters);
            });
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(ek1, stream));
        Counters& done = host;
        HIP_TRY(hipMemcpyAsync(&done, counters, sizeof(Counters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(ev_end, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float ms = 0.f, total = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ek0, ek1));
        HIP_TRY(hipEventElapsedTime(&total, ev_begin, ev_end));
        if (done.paths != rp.n_work)
            return set_error(MRT_ERR_DEVICE, "the kernels ended " + std::to_string(done.paths) + " paths of " + std::to_string(rp.n_work));
        if (stats) {
            stats->paths = done.paths;
            stats->rays = done.rays;
            stats->launches = n_launch;
            stats->kernel_ms = ms;
            stats->total_ms = total;
            stats->segment_kernel_avg_ms = n_launch ? ms / (double)n_launch : 0.0;
        }
        return MRT_OK;
    }
    int cur = 0;
    hipLaunchKernelGGL(k_init_pool, dim3(n_slots / BLOCK), dim3(BLOCK), 0, stream, pools[cur], n_slots, wave_chunks);
    HIP_TRY(hipGetLastError());

    double kernel_ms = 0.0;
    uint64_t launches = 0;
    uint32_t active = n_slots;
    unsigned long long handed_before = 0;
    for (;;) {
        HIP_TRY(hipMemsetAsync(&counters->live, 0, sizeof(unsigned int) * 2 + sizeof(unsigned long long), stream));
        rp.n_slots = active;
        HIP_TRY(hipEventRecord(ek0, stream));
        {
            // (`active` is a multiple of BLOCK, which both workgroup sizes divide)
            dispatch_feat(sc->features, KernelFeats{}, [&](auto feat) {
                constexpr int FT = decltype(feat)::value;
                if (wave_sync)
                    hipLaunchKernelGGL((k_segments<true, FT>), dim3(active / seg_block(true)), dim3(seg_block(true)),
                                       lds_bytes(true, FT, seg_block(true)), stream, view, pools[cur], rp, counters, wave_chunks);
                else
                    hipLaunchKernelGGL((k_segments<false, FT>), dim3(active / seg_block(false)), dim3(seg_block(false)),
                                       lds_bytes(false, FT, seg_block(false)), stream, view, pools[cur], rp, counters, wave_chunks);
            });
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ek1, stream));
        HIP_TRY(hipMemcpyAsync(&host, counters, sizeof(Counters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ek0, ek1));
        kernel_ms += ms;
        launches++;
#ifdef MRT_STAMPS
        std::fprintf(stderr, "[stamps] launch %llu: %.2f ms, %u slots, claim %u, %u live after it, work handed out %llu of %llu\n",
                     (unsigned long long)launches, ms, active, rp.chunk, host.live,
                     (unsigned long long)std::min<unsigned long long>(host.work_next, rp.n_work), (unsigned long long)rp.n_work);
#endif
        {
            // throughput of this launch, for the claim size of the next one (and of the next render)
            const unsigned long long handed = std::min<unsigned long long>(host.work_next, rp.n_work);
            if (handed > handed_before && ms > 0.05f) {
                sc->paths_per_second = (double)(handed - handed_before) / ((double)ms * 1e-3);
                rp.chunk = claim_size(sc->paths_per_second);
            }
            handed_before = handed;
        }
        bool work_left = host.work_next < rp.n_work || host.held > 0;
        rp.all_claimed = host.work_next >= rp.n_work ? 1u : 0u;  // later launches need not ask the counter again
        if (!work_left && host.live == 0) break;
        if (!work_left && can_compact && host.live * 4ull < (unsigned long long)active * 3ull && active > BLOCK) {
            // drain phase: pack the survivors so that later launches cover only live paths
            hipLaunchKernelGGL(k_compact, dim3((active + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, pools[cur],
                               pools[cur ^ 1], active, counters);
            HIP_TRY(hipGetLastError());
            cur ^= 1;
            active = (host.live + BLOCK - 1) / BLOCK * BLOCK;
            if (active > host.live) {
                // slots between live and the block boundary must read as free
                HIP_TRY(hipMemsetAsync(pools[cur].seg() + host.live, 0xFF, sizeof(int32_t) * (active - host.live), stream));
            }
        }
        if (launches > 100000000ull) return set_error(MRT_ERR_DEVICE, "render loop did not terminate");
    }
    HIP_TRY(hipEventRecord(ev_end, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    float total_ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&total_ms, ev_begin, ev_end));
#ifdef MRT_STAMPS
    {
        static const char* names[ST_COUNT] = {"regen", "traverse(all)", "  leaf:solid", "surface+texture", "lambert:generate",
                                              "lambert:light value", "shade(all)", "accumulate", "total", "  leaf:medium", "  regen:claim", "  gen:cosine", "  gen:light", "  gen:sky", "unit vectors (wave)", "  candidate search"};
        std::fprintf(stderr, "[stamps] candidate search per query: %.1f spatial nodes, %.1f marked inner nodes, %.1f marked leaves (%llu queries)\n",
                     (double)host.stamp_counts[0] / (double)std::max(1ull, host.stamp_counts[3]),
                     (double)host.stamp_counts[1] / (double)std::max(1ull, host.stamp_counts[3]),
                     (double)host.stamp_counts[2] / (double)std::max(1ull, host.stamp_counts[3]), host.stamp_counts[3]);
        std::fprintf(stderr, "[stamps] per wave and query: %.1f rounds of the spatial search, %.1f rounds of leaf visits, %.1f lanes asking\n",
                     (double)host.stamp_counts[4] / (double)std::max(1ull, host.stamp_counts[6]),
                     (double)host.stamp_counts[5] / (double)std::max(1ull, host.stamp_counts[6]),
                     (double)host.stamp_counts[3] / (double)std::max(1ull, host.stamp_counts[6]));
        std::fprintf(stderr, "[stamps] queries asked again without the caller's cap: %llu\n", host.stamp_counts[8]);
        std::fprintf(stderr, "[stamps] leaf visits of untransformed static spheres: %.2f per query; rounds of leaf visits without them: %.1f per wave and query\n",
                     (double)host.stamp_counts[11] / (double)std::max(1ull, host.stamp_counts[3]),
                     (double)host.stamp_counts[10] / (double)std::max(1ull, host.stamp_counts[6]));
        {
            unsigned long long od[6] = {0, 0, 0, 0, 0, 0};
            (void)hipMemcpyFromSymbol(od, HIP_SYMBOL(g_oct_diag), sizeof(od));
            if (od[1]) std::fprintf(stderr, "[stamps] octree walk: %llu lane steps in %llu wave rounds (all launches so far): %.1f of 64 lanes stepping\n", od[0], od[1], (double)od[0] / (double)od[1]);
            if (od[1]) std::fprintf(stderr, "[stamps] octree lookups: %llu node loads, %llu lane lookups loaded (%.3f of lane steps), %llu wave lookups loaded (%.3f of rounds), longest chain %.2f loads on average there\n",
                                    od[2], od[3], (double)od[3] / (double)od[0], od[4], (double)od[4] / (double)od[1], (double)od[5] / (double)std::max(1ull, od[4]));
        }
        std::fprintf(stderr, "[stamps] wave-cycles per region (share of total, active lanes of 64)\n");
        for (int i = 0; i < ST_COUNT; ++i)
            std::fprintf(stderr, "[stamps] %-22s %14llu  %5.1f%%  lanes %4.1f\n", names[i], host.stamps[i],
                         100.0 * (double)host.stamps[i] / (double)std::max(1ull, host.stamps[ST_TOTAL]),
                         (double)host.stamp_lanes[i] / (double)std::max(1ull, host.stamps[i]));
    }
#endif
    // every pixel-sample handed out must have ended on the device: a kernel that dropped work is an error,
    // not a fast render
    if (host.paths != rp.n_work)
        return set_error(MRT_ERR_DEVICE, "the kernels ended " + std::to_string(host.paths) + " paths of " + std::to_string(rp.n_work));
    if (stats) {
        stats->paths = host.paths;
        stats->rays = host.rays;
        stats->light_queries = host.light_queries;
        stats->launches = launches;
        stats->kernel_ms = kernel_ms;
        stats->total_ms = total_ms;
        stats->segment_kernel_avg_ms = launches ? kernel_ms / (double)launches : 0.0;
    }
    return MRT_OK;
}

// ---- RCCL, bound at first use (no link-time dependency: single-GPU hosts never load it)
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
    // one communicator set per device list, kept for the life of the process
    std::map<std::vector<int>, std::vector<ncclComm_t>> comms;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.comm_init_all = reinterpret_cast<decltype(r.comm_init_all)>(dlsym(r.lib, "ncclCommInitAll"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(r.lib, "ncclAllReduce"));
        r.group_start = reinterpret_cast<decltype(r.group_start)>(dlsym(r.lib, "ncclGroupStart"));
        r.group_end = reinterpret_cast<decltype(r.group_end)>(dlsym(r.lib, "ncclGroupEnd"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.lib, "ncclGetErrorString"));
        r.ok = r.comm_init_all && r.comm_destroy && r.all_reduce && r.group_start && r.group_end && r.error_string;
    });
    return r;
}
std::mutex g_multi_mu;  // mrt_render_accumulate_multi calls take turns (resident images, communicators)

// In-place sum of one f32 image per device over all devices: one ncclAllReduce per device inside one
// group call (RCCL over xGMI between GPUs; a copy when there is one device).
//   JOIN_DONE         the images hold the sum;
//   JOIN_UNAVAILABLE  RCCL could not be loaded or its communicators could not be made: NO image has been
//                     touched, the caller may sum on the host;
//   JOIN_FAILED       a call failed after the collectives were started: the images may be partly reduced,
//                     nothing can be built on them.
// MRT_MULTI_INJECT_FAILURE=load|collective makes the function answer as if the library were missing / as
// if the collective had failed after it was enqueued (for tests of the two ways out).
enum JoinResult { JOIN_DONE, JOIN_UNAVAILABLE, JOIN_FAILED };
JoinResult all_reduce_images(const std::vector<int>& devices, const std::vector<float*>& images, size_t n_float, std::string& why) {
    const char* inject = std::getenv("MRT_MULTI_INJECT_FAILURE");
    Rccl& r = rccl();
    if (!r.ok || (inject && std::strcmp(inject, "load") == 0)) {
        why = "librccl could not be loaded";
        return JOIN_UNAVAILABLE;
    }
    auto it = r.comms.find(devices);
    if (it == r.comms.end()) {
        std::vector<ncclComm_t> c(devices.size());
        const ncclResult_t rc = r.comm_init_all(c.data(), (int)devices.size(), devices.data());
        if (rc != ncclSuccess) {
            why = std::string("ncclCommInitAll: ") + r.error_string(rc);
            return JOIN_UNAVAILABLE;
        }
        it = r.comms.emplace(devices, std::move(c)).first;
    }
    ncclResult_t rc = r.group_start();
    if (rc != ncclSuccess) {
        why = std::string("ncclGroupStart: ") + r.error_string(rc);
        return JOIN_UNAVAILABLE;  // nothing enqueued yet
    }
    for (size_t g = 0; g < devices.size() && rc == ncclSuccess; ++g) {
        if (hipSetDevice(devices[g]) != hipSuccess) rc = ncclUnhandledCudaError;
        else rc = r.all_reduce(images[g], images[g], n_float, ncclFloat, ncclSum, it->second[g], nullptr);
    }
    const ncclResult_t rc_end = r.group_end();
    if (rc == ncclSuccess) rc = rc_end;
    bool synced = true;
    for (size_t g = 0; g < devices.size(); ++g)
        synced = hipSetDevice(devices[g]) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess && synced;
    if (rc == ncclSuccess && inject && std::strcmp(inject, "collective") == 0) rc = ncclInternalError;
    if (rc != ncclSuccess || !synced) {
        why = rc != ncclSuccess ? std::string("ncclAllReduce: ") + r.error_string(rc) : "synchronisation after the all-reduce failed";
        return JOIN_FAILED;
    }
    return JOIN_DONE;
}

// Body of mrt_render_accumulate_multi (arguments checked by the caller). Each GPU holds ONE resident
// accumulation image (owned by the first handle on it; several handles on a GPU add into it together,
// the adds are atomic). The strip of the host image is uploaded once, to the first GPU; the per-GPU
// images are summed on the devices; the strip is downloaded once.
int render_multi(mrt_scene* const* scenes, uint32_t n_scenes, float* accum_rgb, uint32_t width, uint32_t height,
                 uint32_t x0, uint32_t strip_w, uint32_t spp, uint64_t seed, uint64_t first_sample_index, mrt_stats* stats) {
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (spp == 0) return MRT_OK;
    std::lock_guard<std::mutex> turn(g_multi_mu);
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t n_float = 3 * (size_t)width * height;
    struct Share {
        mrt_stats stats{};
        int rc = MRT_OK;
        std::string message;
        uint32_t spp = 0;
        uint64_t first = 0;
        size_t group = 0;
    };
    std::vector<Share> shares(n_scenes);
    std::vector<int> devices;       // distinct devices, in order of first use
    std::vector<mrt_scene*> owners; // first handle on each device: holds the device's image
    uint64_t next = first_sample_index;
    for (uint32_t i = 0; i < n_scenes; ++i) {
        shares[i].spp = spp / n_scenes + (i < spp % n_scenes ? 1u : 0u);
        shares[i].first = next;
        next += shares[i].spp;
        size_t g = std::find(devices.begin(), devices.end(), scenes[i]->device) - devices.begin();
        if (g == devices.size()) devices.push_back(scenes[i]->device), owners.push_back(scenes[i]);
        shares[i].group = g;
    }
    std::vector<float*> images(devices.size(), nullptr);
    const size_t host_pitch = sizeof(float) * 3 * (size_t)width, strip_bytes = sizeof(float) * 3 * (size_t)strip_w;
    for (size_t g = 0; g < devices.size(); ++g) {
        mrt_scene* o = owners[g];
        HIP_TRY(hipSetDevice(devices[g]));
        if (o->multi_accum_floats != n_float) {
            if (o->multi_accum) (void)hipFree(o->multi_accum);
            o->multi_accum = nullptr, o->multi_accum_floats = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&o->multi_accum), sizeof(float) * n_float));
            o->multi_accum_floats = n_float;
        }
        images[g] = o->multi_accum;
        HIP_TRY(hipMemset(images[g], 0, sizeof(float) * n_float));
        if (g == 0)  // the host image's strip joins the sum through the first device
            HIP_TRY(hipMemcpy2D(images[0] + 3 * (size_t)x0, host_pitch, accum_rgb + 3 * (size_t)x0, host_pitch, strip_bytes, height,
                                hipMemcpyHostToDevice));
    }
    {
        std::vector<std::thread> workers;
        struct Joiner {
            std::vector<std::thread>& w;
            ~Joiner() {
                for (std::thread& t : w)
                    if (t.joinable()) t.join();
            }
        } joiner{workers};
        workers.reserve(n_scenes);
        for (uint32_t i = 0; i < n_scenes; ++i) {
            if (shares[i].spp == 0) continue;
            workers.emplace_back([&, i] {
                Share& s = shares[i];
                try {
                    s.rc = render_device(scenes[i], images[s.group], width, height, x0, strip_w, s.spp, seed, s.first, nullptr, &s.stats);
                    if (s.rc != MRT_OK) s.message = g_last_error;  // the worker thread's message
                } catch (const std::exception& e) {
                    s.rc = MRT_ERR_OUT_OF_MEMORY, s.message = e.what();
                }
            });
        }
    }
    for (const Share& s : shares)
        if (s.rc != MRT_OK) return set_error(s.rc, s.message);
    // join the per-device images: on the devices when RCCL is there (always for several devices; for one
    // device only when MRT_MULTI_FORCE_RCCL is set, which lets a one-GPU host exercise the collective)
    const bool forced = std::getenv("MRT_MULTI_FORCE_RCCL") != nullptr;
    const bool want_collective = devices.size() > 1 || forced;
    bool reduced = devices.size() == 1;
    std::string why;
    if (want_collective) {
        const JoinResult jr = all_reduce_images(devices, images, n_float, why);
        // a failure after the collectives were enqueued leaves partly reduced images: adding the other devices'
        // strips on the host would count samples twice. Only "RCCL is not there" falls back to the host sum.
        if (jr == JOIN_FAILED) return set_error(MRT_ERR_DEVICE, why);
        if (jr == JOIN_UNAVAILABLE && forced && std::getenv("MRT_MULTI_INJECT_FAILURE") == nullptr) return set_error(MRT_ERR_DEVICE, why);
        reduced = jr == JOIN_DONE || devices.size() == 1;
    }
    HIP_TRY(hipSetDevice(devices[0]));
    HIP_TRY(hipMemcpy2D(accum_rgb + 3 * (size_t)x0, host_pitch, images[0] + 3 * (size_t)x0, host_pitch, strip_bytes, height,
                        hipMemcpyDeviceToHost));
    if (!reduced) {
        // no collective: the other devices' strips are added on the host, in device order
        std::vector<float> tmp((size_t)strip_w * 3 * height);
        for (size_t g = 1; g < devices.size(); ++g) {
            HIP_TRY(hipSetDevice(devices[g]));
            HIP_TRY(hipMemcpy2D(tmp.data(), strip_bytes, images[g] + 3 * (size_t)x0, host_pitch, strip_bytes, height, hipMemcpyDeviceToHost));
            for (uint32_t y = 0; y < height; ++y) {
                float* dst = accum_rgb + ((size_t)y * width + x0) * 3;
                const float* src = tmp.data() + (size_t)y * strip_w * 3;
                for (size_t k = 0; k < (size_t)strip_w * 3; ++k) dst[k] += src[k];
            }
        }
    }
    mrt_stats total{};
    for (const Share& s : shares) {
        if (s.spp == 0) continue;
        total.paths += s.stats.paths, total.rays += s.stats.rays, total.light_queries += s.stats.light_queries;
        total.launches += s.stats.launches;
        total.kernel_ms = std::max(total.kernel_ms, s.stats.kernel_ms);
        total.segment_kernel_avg_ms = std::max(total.segment_kernel_avg_ms, s.stats.segment_kernel_avg_ms);
    }
    total.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (stats) *stats = total;
    return MRT_OK;
}

}  // namespace

extern "C" {

int mrt_abi_version(void) { return MRT_ABI_VERSION; }

int mrt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* mrt_last_error(void) { return g_last_error.c_str(); }

mrt_scene* mrt_scene_create(const mrt_scene_desc* desc, int device, char* err, size_t errlen) {
    auto report = [&](const std::string& msg) {
        g_last_error = msg;
        if (err && errlen) std::snprintf(err, errlen, "%s", msg.c_str());
    };
    std::unique_ptr<mrt_scene> sc(new (std::nothrow) mrt_scene());
    if (!sc) {
        report("out of host memory");
        return nullptr;
    }
    std::string why;
    int rc;
    try {
        rc = flatten_scene(desc, &sc->flat, &why);
    } catch (const std::exception& e) {
        rc = MRT_ERR_OUT_OF_MEMORY;
        why = e.what();
    }
    if (rc != MRT_OK) {
        report(why);
        return nullptr;
    }
    sc->device = device;
    sc->features = scene_features(sc->flat);
    if (device >= 0) {
        int n = mrt_device_count();
        if (n <= 0) {
            report("no HIP device is visible; the HIP path is required (there is no CPU fallback)");
            return nullptr;
        }
        if (device >= n) {
            report("device ordinal out of range");
            return nullptr;
        }
        if (hipSetDevice(device) != hipSuccess) {
            report("hipSetDevice failed");
            return nullptr;
        }
        if (upload_scene(sc.get()) != MRT_OK || create_call_resources(sc.get()) != MRT_OK) {
            report(g_last_error);
            return nullptr;
        }
        // set-up step of the no-allocation entry point: this device's address of its scratch array
        // (a failure is reported by that entry point, not here)
        (void)resolve_scratch(device);
    }
    return sc.release();
}

void mrt_scene_destroy(mrt_scene* scene) {
    if (!scene) return;
    if (scene->device >= 0) (void)hipSetDevice(scene->device);
    delete scene;
}

int mrt_scene_bvh(const mrt_scene* scene, mrt_bvh_node* out, size_t capacity, size_t* n_nodes) {
    if (!scene) return set_error(MRT_ERR_INVALID_ARGUMENT, "scene is NULL");
    const auto& v = scene->flat.bvh_full;
    if (n_nodes) *n_nodes = v.size();
    if (out) {
        if (capacity < v.size()) return set_error(MRT_ERR_INVALID_ARGUMENT, "capacity too small");
        if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(mrt_bvh_node));
    }
    return MRT_OK;
}

int mrt_scene_set_option(mrt_scene* scene, const char* name, int64_t value) {
    if (!scene || !name) return set_error(MRT_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(scene->mu);
    std::string n(name);
    if (n == "pool_slots") {
        if (value < BLOCK) return set_error(MRT_ERR_INVALID_ARGUMENT, "pool_slots too small");
        scene->opt_pool_slots = value;
    } else if (n == "segments_per_launch") {
        if (value < 1 || value > 65536) return set_error(MRT_ERR_INVALID_ARGUMENT, "segments_per_launch must be 1..65536");
        scene->opt_steps = value;
    } else if (n == "trace_zero_throughput") {
        scene->opt_trace_zero = value ? 1 : 0;
    } else if (n == "compaction") {
        scene->opt_compact = value ? 1 : 0;
    } else if (n == "shader") {
        if (value < MRT_SHADER_RAYTRACING || value > MRT_SHADER_LIGHT_MAP)
            return set_error(MRT_ERR_INVALID_ARGUMENT, "shader must be one of MRT_SHADER_*");
        scene->opt_shader = value;
    } else if (n == "traversal") {
        if (value < 0 || value > 4) return set_error(MRT_ERR_INVALID_ARGUMENT, "traversal must be 0..4");
        scene->opt_traversal = value;
    } else if (n == "octree_suspend") {
        scene->opt_oct_suspend = value ? 1 : 0;
        if (scene->device < 0) return MRT_OK;
        HIP_TRY(hipSetDevice(scene->device));
        return upload_views(scene);
    } else if (n == "candidate_queue") {
        if (value < 256 || value > CAND_QUEUE) return set_error(MRT_ERR_INVALID_ARGUMENT, "candidate_queue must be 256..2048");
        if (scene->device < 0) return set_error(MRT_ERR_INVALID_ARGUMENT, "scene has no device image");
        HIP_TRY(hipSetDevice(scene->device));
        scene->view.cand_queue = (int32_t)value;
        return upload_views(scene);
    } else {
        return set_error(MRT_ERR_INVALID_ARGUMENT, "unknown option: " + n);
    }
    return MRT_OK;
}

int mrt_camera_rays(mrt_scene* scene, uint32_t width, uint32_t height, const int32_t* pixels, size_t n,
                    uint64_t seed, uint64_t sample_index, double* rays_out) {
    if (!scene || scene->device < 0) return set_error(MRT_ERR_INVALID_ARGUMENT, "scene has no device image");
    if (!pixels || !rays_out) return set_error(MRT_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (width < 2 || height < 2 || n > 0x7FFFFFFFu) return set_error(MRT_ERR_INVALID_ARGUMENT, "bad size");
    for (size_t i = 0; i < n; ++i)
        if (pixels[2 * i] < 0 || (uint32_t)pixels[2 * i] >= width || pixels[2 * i + 1] < 0 ||
            (uint32_t)pixels[2 * i + 1] >= height)
            return set_error(MRT_ERR_INVALID_ARGUMENT, "pixel outside the image");
    if (n == 0) return MRT_OK;
    std::lock_guard<std::mutex> lock(scene->mu);
    HIP_TRY(hipSetDevice(scene->device));
    DeviceBuffers tmp;
    int32_t* dpix;
    double* dout;
    int rc;
    if ((rc = tmp.alloc(sizeof(int32_t) * 2 * n, (void**)&dpix)) != MRT_OK) return rc;
    if ((rc = tmp.alloc(sizeof(double) * 7 * n, (void**)&dout)) != MRT_OK) return rc;
    HIP_TRY(hipMemcpy(dpix, pixels, sizeof(int32_t) * 2 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_camera_rays, dim3(((uint32_t)n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0,
                       device_view(scene, traversal_mode(scene), false), width,
                       height, dpix, (uint32_t)n, seed, sample_index, dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rays_out, dout, sizeof(double) * 7 * n, hipMemcpyDeviceToHost));
    return MRT_OK;
}

int mrt_trace_rays(mrt_scene* scene, const double* rays, size_t n, uint64_t seed, int32_t* obj_id, double* t,
                   double* normal, double* position, int32_t* front_face) {
    if (!scene || scene->device < 0) return set_error(MRT_ERR_INVALID_ARGUMENT, "scene has no device image");
    if (!rays && n) return set_error(MRT_ERR_INVALID_ARGUMENT, "rays is NULL");
    if (n > 0x7FFFFFFFu) return set_error(MRT_ERR_INVALID_ARGUMENT, "too many rays");
    if (n == 0) return MRT_OK;
    std::lock_guard<std::mutex> lock(scene->mu);
    HIP_TRY(hipSetDevice(scene->device));
    DeviceBuffers tmp;
    double *drays, *dt = nullptr, *dn = nullptr, *dp = nullptr;
    int32_t *did = nullptr, *dff = nullptr;
    int rc;
    if ((rc = tmp.alloc(sizeof(double) * 7 * n, (void**)&drays)) != MRT_OK) return rc;
    if (obj_id && (rc = tmp.alloc(sizeof(int32_t) * n, (void**)&did)) != MRT_OK) return rc;
    if (t && (rc = tmp.alloc(sizeof(double) * n, (void**)&dt)) != MRT_OK) return rc;
    if (normal && (rc = tmp.alloc(sizeof(double) * 3 * n, (void**)&dn)) != MRT_OK) return rc;
    if (position && (rc = tmp.alloc(sizeof(double) * 3 * n, (void**)&dp)) != MRT_OK) return rc;
    if (front_face && (rc = tmp.alloc(sizeof(int32_t) * n, (void**)&dff)) != MRT_OK) return rc;
    HIP_TRY(hipMemcpy(drays, rays, sizeof(double) * 7 * n, hipMemcpyHostToDevice));
    {
        const dim3 grid(((uint32_t)n + BLOCK - 1) / BLOCK), block(BLOCK);
        const int trav = traversal_mode(scene);
        const bool ws = trav != TRAV_LANE && trav != TRAV_CANDIDATES;
        const SceneView* view = device_view(scene, trav, false);
        dispatch_feat(scene->features, KernelFeats{}, [&](auto feat) {
            constexpr int FT = decltype(feat)::value;
            if (ws)
                hipLaunchKernelGGL((k_trace_rays<true, FT>), grid, block, lds_bytes(true, FT, BLOCK), 0, view, drays, (uint32_t)n, seed, did, dt, dn, dp, dff);
            else
                hipLaunchKernelGGL((k_trace_rays<false, FT>), grid, block, lds_bytes(false, FT, BLOCK), 0, view, drays, (uint32_t)n, seed, did, dt, dn, dp, dff);
        });
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (obj_id) HIP_TRY(hipMemcpy(obj_id, did, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (t) HIP_TRY(hipMemcpy(t, dt, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (normal) HIP_TRY(hipMemcpy(normal, dn, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
    if (position) HIP_TRY(hipMemcpy(position, dp, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
    if (front_face) HIP_TRY(hipMemcpy(front_face, dff, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return MRT_OK;
}

int mrt_render_accumulate_device(mrt_scene* scene, float* accum_rgb_device, uint32_t width, uint32_t height,
                                 uint32_t x0, uint32_t strip_w, uint32_t spp, uint64_t seed,
                                 uint64_t first_sample_index, void* stream, mrt_stats* stats) {
    return render_device(scene, accum_rgb_device, width, height, x0, strip_w, spp, seed, first_sample_index,
                         static_cast<hipStream_t>(stream), stats);
}

int mrt_render_accumulate_device_noalloc(mrt_scene* scene, float* accum_rgb_device, uint32_t width, uint32_t height,
                                         uint32_t x0, uint32_t strip_w, uint32_t spp, uint64_t seed,
                                         uint64_t first_sample_index, void* stream, mrt_stats* stats) {
    if (!scene || scene->device < 0) return set_error(MRT_ERR_INVALID_ARGUMENT, "scene has no device image");
    if (scene->device >= MAX_DEVICES) return set_error(MRT_ERR_UNSUPPORTED, "device ordinal beyond the scratch table");
    HIP_TRY(hipSetDevice(scene->device));
    const hipError_t e = resolve_scratch(scene->device);  // resolved when the scene was created
    if (e != hipSuccess)
        return set_error(MRT_ERR_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    ScratchDevice& dev = g_scratch_dev[scene->device];
    std::lock_guard<std::mutex> turn(dev.in_use);  // one call at a time per GPU: they share its array
    return render_device(scene, accum_rgb_device, width, height, x0, strip_w, spp, seed, first_sample_index,
                         static_cast<hipStream_t>(stream), stats, dev.ptr);
}

int mrt_render_accumulate(mrt_scene* scene, float* accum_rgb, uint32_t width, uint32_t height, uint32_t x0,
                          uint32_t strip_w, uint32_t spp, uint64_t seed, uint64_t first_sample_index,
                          mrt_stats* stats) {
    if (!scene || scene->device < 0) return set_error(MRT_ERR_INVALID_ARGUMENT, "scene has no device image");
    if (!accum_rgb) return set_error(MRT_ERR_INVALID_ARGUMENT, "accumulation buffer is NULL");
    if (width < 2 || height < 2 || (uint64_t)width * height > 0xFFFFFFFFull / 3)
        return set_error(MRT_ERR_INVALID_ARGUMENT, "bad image size");
    HIP_TRY(hipSetDevice(scene->device));
    size_t bytes = sizeof(float) * 3 * (size_t)width * height;
    DeviceBuffers tmp;
    float* dacc;
    int rc;
    if ((rc = tmp.alloc(bytes, (void**)&dacc)) != MRT_OK) return rc;
    HIP_TRY(hipMemcpy(dacc, accum_rgb, bytes, hipMemcpyHostToDevice));
    rc = render_device(scene, dacc, width, height, x0, strip_w, spp, seed, first_sample_index, nullptr, stats);
    if (rc != MRT_OK) return rc;
    HIP_TRY(hipMemcpy(accum_rgb, dacc, bytes, hipMemcpyDeviceToHost));
    return MRT_OK;
}

int mrt_render_accumulate_multi(mrt_scene* const* scenes, uint32_t n_scenes, float* accum_rgb, uint32_t width,
                                uint32_t height, uint32_t x0, uint32_t strip_w, uint32_t spp, uint64_t seed,
                                uint64_t first_sample_index, mrt_stats* stats) {
    if (!scenes || n_scenes == 0) return set_error(MRT_ERR_INVALID_ARGUMENT, "no scene handles");
    for (uint32_t i = 0; i < n_scenes; ++i)
        if (!scenes[i] || scenes[i]->device < 0) return set_error(MRT_ERR_INVALID_ARGUMENT, "scene has no device image");
    if (!accum_rgb) return set_error(MRT_ERR_INVALID_ARGUMENT, "accumulation buffer is NULL");
    if (width < 2 || height < 2 || (uint64_t)width * height > 0xFFFFFFFFull / 3)
        return set_error(MRT_ERR_INVALID_ARGUMENT, "bad image size");
    if (x0 >= width || strip_w == 0 || strip_w > width - x0) return set_error(MRT_ERR_INVALID_ARGUMENT, "strip outside the image");
    try {
        return render_multi(scenes, n_scenes, accum_rgb, width, height, x0, strip_w, spp, seed, first_sample_index, stats);
    } catch (const std::bad_alloc&) {
        return set_error(MRT_ERR_OUT_OF_MEMORY, "out of host memory");
    } catch (const std::exception& e) {
        return set_error(MRT_ERR_DEVICE, e.what());
    }
}

int mrt_to_image(const float* accum_rgb, uint32_t width, uint32_t height, uint32_t num_samples, uint8_t* rgb8_out) {
    if (!accum_rgb || !rgb8_out || num_samples == 0) return set_error(MRT_ERR_INVALID_ARGUMENT, "bad argument");
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            const float* p = accum_rgb + ((size_t)x + (size_t)y * width) * 3;
            uint8_t* o = rgb8_out + ((size_t)x + (size_t)(height - y - 1) * width) * 3;
            for (int c = 0; c < 3; ++c) {
                float s = std::sqrt(p[c] / (float)num_samples);  // rgb_color.rs:75-80
                if (s < 0.0f) s = 0.0f;
                else if (s > 1.0f) s = 1.0f;
                float r = std::round(s * 255.0f);
                o[c] = std::isnan(r) ? (uint8_t)0 : (uint8_t)r;
            }
        }
    return MRT_OK;
}

}  // extern "C"
