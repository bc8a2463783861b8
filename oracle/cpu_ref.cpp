// cpu_ref.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (C++17, f64 geometry / f32 colour) of the reference's
// per-pixel path-tracing loop, written from the cited source lines of
// scifi6546/ray_tracing (paths relative to the reference checkout). It keeps
// the reference's structure: recursive `ray_color`, boxed BVH tree walked
// recursively, leaf bounding boxes (and matrix inverses) recomputed on every
// visit, materials evaluated eagerly for every candidate hit, linear light
// list.  It is the checker for the HIP path and the "port" CPU baseline; the
// shipped library never links or calls it.
//
// Parity status: the reference is Rust and cannot be built in this
// environment (no cargo/rustc, crates not vendored), and its own unit tests
// hold no vectors for intersection, scattering or radiance.  For those parts
// this file is "parity unpinned": it is pinned only by being a line-by-line
// restatement.  The octree / arena / colour-hex vectors that the reference's
// tests do hold are pinned in octree_ref.cpp.
//
// The only deliberate deviation from the reference: `rand::random()`
// (cpu_rt/src/prelude.rs:82-87, a thread-local OS-seeded ChaCha) is replaced
// by the counter-based generator specified in include/mrt.h, addressed by
// (path key, segment, dimension), or (rng_mode 1) by a sequential
// std::mt19937_64 used as a statistically independent estimator.

#include "../include/mrt.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <optional>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

#include "octree_ref.h"

namespace {

// ---------------------------------------------------------------- math (cgmath 0.18)
struct V3 {
    double x, y, z;
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double magnitude(V3 a) { return std::sqrt(dot(a, a)); }
// cgmath InnerSpace::normalize = self * (1 / magnitude)
inline V3 normalize(V3 a) { return a * (1.0 / magnitude(a)); }

struct V4 {
    double x, y, z, w;
};
struct M3 {
    V3 c[3];  // columns
};
struct M4 {
    V4 c[4];  // columns
};
inline V4 mul(const M4& m, V4 v) {
    // cgmath: self[0]*v.x + self[1]*v.y + self[2]*v.z + self[3]*v.w
    return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z + m.c[3].x * v.w,
            m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z + m.c[3].y * v.w,
            m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z + m.c[3].z * v.w,
            m.c[0].w * v.x + m.c[1].w * v.y + m.c[2].w * v.z + m.c[3].w * v.w};
}
inline V3 mul(const M3& m, V3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }
inline double det3(const M3& m) {
    return m.c[0].x * (m.c[1].y * m.c[2].z - m.c[2].y * m.c[1].z) -
           m.c[1].x * (m.c[0].y * m.c[2].z - m.c[2].y * m.c[0].z) +
           m.c[2].x * (m.c[0].y * m.c[1].z - m.c[1].y * m.c[0].z);
}
inline M3 transpose(const M3& m) {
    return {{{m.c[0].x, m.c[1].x, m.c[2].x},
             {m.c[0].y, m.c[1].y, m.c[2].y},
             {m.c[0].z, m.c[1].z, m.c[2].z}}};
}
inline bool invert3(const M3& m, M3* out) {
    double det = det3(m);
    if (det == 0.0) return false;
    M3 r{{cross(m.c[1], m.c[2]) / det, cross(m.c[2], m.c[0]) / det, cross(m.c[0], m.c[1]) / det}};
    *out = transpose(r);
    return true;
}
inline double at4(const M4& m, int col, int row) {
    const V4& v = m.c[col];
    return row == 0 ? v.x : (row == 1 ? v.y : (row == 2 ? v.z : v.w));
}
// 3x3 minor of m without column `col` and row `row`
inline double minor4(const M4& m, int col, int row) {
    double a[3][3];
    int ci = 0;
    for (int c = 0; c < 4; ++c) {
        if (c == col) continue;
        int ri = 0;
        for (int r = 0; r < 4; ++r) {
            if (r == row) continue;
            a[ci][ri] = at4(m, c, r);
            ++ri;
        }
        ++ci;
    }
    M3 s{{{a[0][0], a[0][1], a[0][2]}, {a[1][0], a[1][1], a[1][2]}, {a[2][0], a[2][1], a[2][2]}}};
    return det3(s);
}
inline double det4(const M4& m) {
    double d = 0.0;
    for (int c = 0; c < 4; ++c) {
        double sign = (c & 1) ? -1.0 : 1.0;
        d += sign * at4(m, c, 0) * minor4(m, c, 0);
    }
    return d;
}
inline bool invert4(const M4& m, M4* out) {
    double det = det4(m);
    if (det == 0.0) return false;
    double inv_det = 1.0 / det;
    double r[4][4];  // r[col][row] of the inverse
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) {
            // inverse(row, col) = cofactor(col, row) / det
            double sign = ((col + row) & 1) ? -1.0 : 1.0;
            r[col][row] = minor4(m, row, col) * sign * inv_det;
        }
    for (int c = 0; c < 4; ++c) out->c[c] = {r[c][0], r[c][1], r[c][2], r[c][3]};
    return true;
}

// prelude/rgb_color.rs:4-8
struct Rgb {
    float r, g, b;
};
inline Rgb operator*(Rgb a, Rgb b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Rgb operator+(Rgb a, Rgb b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Rgb mul_f64(Rgb a, double s) {  // rgb_color.rs:138-147
    float f = (float)s;
    return {a.r * f, a.g * f, a.b * f};
}
inline Rgb div_f64(Rgb a, double s) {  // rgb_color.rs:170-178
    float f = (float)s;
    return {a.r / f, a.g / f, a.b / f};
}
const Rgb BLACK{0.f, 0.f, 0.f};

// prelude.rs:155-165
struct Ray {
    V3 origin;
    V3 direction;
    double time;
    V3 at(double t) const { return origin + t * direction; }
};

// ---------------------------------------------------------------- random numbers
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline uint64_t path_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
    return mix64(mix64(seed + 0x9E3779B97F4A7C15ULL * (pixel + 1)) ^
                 (0xD1B54A32D192ED03ULL * (sample + 1)));
}
struct RngCtx {
    uint64_t key = 0;
    uint32_t seg = 0;
    uint32_t medium_ordinal = 0;
    uint32_t oct_ordinal = 0;  // volume entries of the octree walks of the current nearest_hit
    bool in_light_query = false;
    int mode = 0;  // 0 counter-based, 1 sequential mt19937_64
    std::mt19937_64 mt;
    uint64_t rays = 0, light_queries = 0;
};
thread_local RngCtx g_rng;
// Diagnostic: when set, every World::nearest_hit of the calling thread appends the number of octree walk steps it took.
thread_local std::vector<uint32_t>* g_walk_steps = nullptr;

inline uint64_t rand_u64(uint32_t dim) {
    if (g_rng.mode == 1) return g_rng.mt();
    return mix64(g_rng.key + 0x9E3779B97F4A7C15ULL * ((((uint64_t)g_rng.seg << 16) | dim) + 1));
}
inline double rand_f64(uint32_t dim) { return (double)(rand_u64(dim) >> 11) * 0x1.0p-53; }
// prelude.rs:82-84
inline double rand_scalar(double min, double max, uint32_t dim) {
    return rand_f64(dim) * (max - min) + min;
}
// prelude.rs:85-87
inline uint32_t rand_u32(uint32_t min, uint32_t max, uint32_t dim) {
    return ((uint32_t)(rand_u64(dim) >> 32) % (max - min)) + min;
}
// ray_tracer.rs:48-63 (rejection in the unit ball; NOT normalised)
inline V3 rand_unit_vec(uint32_t base) {
    for (uint32_t k = 0; k < MRT_MAX_REJECT; ++k) {
        V3 rv{rand_f64(base + 3 * k), rand_f64(base + 3 * k + 1), rand_f64(base + 3 * k + 2)};
        V3 v = 2.0 * (rv - V3{0.5, 0.5, 0.5});
        if (dot(v, v) < 1.0) return v;
    }
    return {0.0, 0.0, 0.0};
}

// prelude.rs:103-130
struct OrthoNormalBasis {
    V3 axis[3];
    static OrthoNormalBasis build_from_w(V3 n) {
        V3 w = normalize(n);
        V3 a = std::fabs(w.x) > 0.9 ? V3{0.0, 1.0, 0.0} : V3{1.0, 0.0, 0.0};
        V3 v = normalize(cross(w, a));
        V3 u = cross(w, v);
        return {{u, v, w}};
    }
    V3 local(V3 a) const { return a.x * axis[0] + a.y * axis[1] + a.z * axis[2]; }
    V3 w() const { return axis[2]; }
};
// prelude.rs:145-153
inline V3 random_cosine_direction() {
    double r1 = rand_scalar(0.0, 1.0, MRT_DIM_COS_R1);
    double r2 = rand_scalar(0.0, 1.0, MRT_DIM_COS_R2);
    double z = std::sqrt(1.0 - r2);
    double phi = 2.0 * M_PI * r1;
    double x = std::cos(phi) * std::sqrt(r2);
    double y = std::sin(phi) * std::sqrt(r2);
    return {x, y, z};
}
// lib.rs:8-10
inline V3 reflect(V3 v, V3 normal) { return v - 2.0 * dot(v, normal) * normal; }

// ---------------------------------------------------------------- textures (texture.rs)
struct Image {
    int width, height;
    std::vector<float> rgb;
    Rgb get_xy(size_t x, size_t y) const {
        const float* p = &rgb[(x + y * (size_t)width) * 3];
        return {p[0], p[1], p[2]};
    }
};
inline size_t f64_as_usize(double v) {  // Rust `as usize`: saturating, NaN -> 0
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return (size_t)-1;
    return (size_t)v;
}
struct Texture {
    virtual ~Texture() = default;
    virtual Rgb color(double u, double v, V3 pos) const = 0;
};
struct SolidColor : Texture {  // texture.rs:33-44
    Rgb c;
    Rgb color(double, double, V3) const override { return c; }
};
struct ImageTexture : Texture {  // texture.rs:217-224, parallel_image.rs:66-71
    const Image* img;
    Rgb color(double u, double v, V3) const override {
        size_t w = (size_t)img->width, h = (size_t)img->height;
        size_t x = std::min(f64_as_usize(u * ((double)w - 1.0)), w - 1);
        double vv = 1.0 - v;
        size_t y = std::min(f64_as_usize(vv * ((double)h - 1.0)), h - 1);
        return img->get_xy(x, y);
    }
};
struct MultiplyTexture : Texture {  // texture.rs:12-32
    const Texture *a, *b;
    Rgb color(double u, double v, V3 p) const override { return a->color(u, v, p) * b->color(u, v, p); }
};
struct CheckerTexture : Texture {  // texture.rs:46-70
    const Texture *odd, *even;
    Rgb color(double u, double v, V3 p) const override {
        double s = std::sin(10.0 * p.x) * std::sin(10.0 * p.y) * std::sin(10.0 * p.z);
        return s < 0.0 ? odd->color(u, v, p) : even->color(u, v, p);
    }
};
inline int f64_as_i32(double v) {  // Rust `as i32`: toward zero, saturating, NaN -> 0
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)v;
}
struct PerlinTexture : Texture {  // texture.rs:71-190; the random tables are scene data
    const mrt_perlin* pn;
    double noise(V3 point) const {  // :124-149 with trilinear_interp :95-122
        double u = point.x - std::floor(point.x), v = point.y - std::floor(point.y), w = point.z - std::floor(point.z);
        int i = f64_as_i32(std::floor(point.x)), j = f64_as_i32(std::floor(point.y)), k = f64_as_i32(std::floor(point.z));
        V3 c[2][2][2];
        for (int di = 0; di < 2; ++di)
            for (int dj = 0; dj < 2; ++dj)
                for (int dk = 0; dk < 2; ++dk) {
                    uint32_t idx = pn->perm_x[(i + di) & 255] ^ pn->perm_y[(j + dj) & 255] ^ pn->perm_z[(k + dk) & 255];
                    c[di][dj][dk] = V3{pn->ran_float[idx][0], pn->ran_float[idx][1], pn->ran_float[idx][2]};
                }
        double uu = u * u * (3.0 - 2.0 * u), vv = v * v * (3.0 - 2.0 * v), ww = w * w * (3.0 - 2.0 * w);
        double accum = 0.0;
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int cc = 0; cc < 2; ++cc) {
                    V3 weight_v{u - (double)a, v - (double)b, w - (double)cc};
                    accum += ((double)a * uu + (double)(1 - a) * (1.0 - uu)) * ((double)b * vv + (double)(1 - b) * (1.0 - vv)) *
                             ((double)cc * ww + (double)(1 - cc) * (1.0 - ww)) * dot(c[a][b][cc], weight_v);
                }
        return accum;
    }
    double turbulence(V3 point, uint32_t depth) const {  // :150-160
        double acum = 0.0, weight = 1.0;
        V3 temp_point = point;
        for (uint32_t n = 0; n < depth; ++n) {
            acum += weight * noise(temp_point);
            weight *= 0.5;
            temp_point = temp_point * 2.0;
        }
        return std::fabs(acum);
    }
    Rgb color(double, double, V3 pos) const override {  // :185-189
        double f = turbulence(pos * 2.0, 7);
        return {(float)f, (float)f, (float)f};
    }
};
struct DebugV : Texture {  // texture.rs:225-239
    Rgb color(double, double v, V3) const override {
        if (!std::isnan(v)) return {(float)v, (float)v, (float)v};
        return BLACK;
    }
};

// ---------------------------------------------------------------- hit records (hittable.rs:323-364)
struct HitRay {
    V3 position, direction, normal;
    bool front_face;
    double u, v;
};
enum class PdfKind { None, Lambertian, Isotropic };
enum class SpdfKind { Panic, Lambertian, VoxelVolume };
struct ScatterRecord {  // pdf.rs:284-290
    std::optional<Ray> specular_ray;
    Rgb attenuation;
    PdfKind pdf = PdfKind::None;
    V3 pdf_normal{0, 0, 0};
    SpdfKind scattering_pdf = SpdfKind::Panic;
};
enum class EffectKind { Scatter, Emmit, NoEmmit };
struct MaterialEffect {
    EffectKind kind = EffectKind::NoEmmit;
    ScatterRecord scatter;
    Rgb emitted{0, 0, 0};
};
struct HitRecord {
    V3 position, normal;
    double t;
    bool front_face;
    double u, v;
    MaterialEffect material_effect;
    int object = -1;  // annotation only: index of the BVH leaf object that produced the record
};

// ---------------------------------------------------------------- materials (material.rs)
struct Material {
    virtual ~Material() = default;
    virtual std::optional<ScatterRecord> scatter(const Ray& ray_in, const HitRay& rec) const = 0;
    virtual std::optional<Rgb> emmit(const HitRay&) const { return std::nullopt; }
};
struct Lambertian : Material {  // material.rs:26-78
    const Texture* albedo;
    std::optional<ScatterRecord> scatter(const Ray&, const HitRay& rec) const override {
        ScatterRecord s;
        s.attenuation = albedo->color(rec.u, rec.v, rec.position);
        s.pdf = PdfKind::Lambertian;
        s.pdf_normal = rec.normal;
        s.scattering_pdf = SpdfKind::Lambertian;
        return s;
    }
};
struct Metal : Material {  // material.rs:80-134
    const Texture* albedo;
    double fuzz;
    std::optional<ScatterRecord> scatter(const Ray& ray_in, const HitRay& rec) const override {
        V3 reflected = reflect(normalize(ray_in.direction), rec.normal);
        if (dot(reflected, rec.normal) > 0.0) {
            ScatterRecord s;
            s.specular_ray = Ray{rec.position, reflected + fuzz * rand_unit_vec(MRT_DIM_METAL_FUZZ),
                                 ray_in.time};
            s.attenuation = albedo->color(rec.u, rec.v, rec.position);
            return s;
        }
        return std::nullopt;
    }
};
struct Dielectric : Material {  // material.rs:135-206
    double index_refraction;
    Rgb color;
    static V3 refract(V3 uv, V3 n, double etai_over_etat) {
        double cos_theta = std::fmin(dot(n, -1.0 * uv), 1.0);
        V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
        V3 r_out_parallel = -1.0 * n * std::sqrt(std::fabs(1.0 - dot(r_out_perp, r_out_perp)));
        return r_out_perp + r_out_parallel;
    }
    static double powi(double b, int e) {
        double r = 1.0;
        for (int i = 0; i < e; ++i) r *= b;
        return r;
    }
    static double reflectance(double cosine, double ref_idx) {
        double r0 = powi((1.0 - ref_idx) / (1.0 + ref_idx), 2);
        return r0 + (1.0 - r0) * powi(1.0 - cosine, 5);
    }
    std::optional<ScatterRecord> scatter(const Ray& ray_in, const HitRay& rec) const override {
        double refraction_ratio = rec.front_face ? 1.0 / index_refraction : index_refraction;
        V3 unit_direction = normalize(ray_in.direction);
        double cos_theta = std::fmin(dot(rec.normal, -1.0 * unit_direction), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool can_not_refract = (refraction_ratio * sin_theta) > 1.0;
        V3 direction;
        if (can_not_refract ||
            reflectance(cos_theta, refraction_ratio) > rand_f64(MRT_DIM_DIELECTRIC))
            direction = reflect(unit_direction, rec.normal);
        else
            direction = refract(unit_direction, rec.normal, refraction_ratio);
        ScatterRecord s;
        s.specular_ray = Ray{rec.position, direction, ray_in.time};
        s.attenuation = color;
        return s;
    }
};
struct DiffuseLight : Material {  // material.rs:208-249
    const Texture* emit;
    std::optional<ScatterRecord> scatter(const Ray&, const HitRay&) const override {
        return std::nullopt;
    }
    std::optional<Rgb> emmit(const HitRay& rec) const override {
        if (rec.front_face) return emit->color(rec.u, rec.v, rec.position);
        return std::nullopt;
    }
};
struct Isotropic : Material {  // material.rs:251-295
    const Texture* albedo;
    std::optional<ScatterRecord> scatter(const Ray& ray_in, const HitRay& rec) const override {
        ScatterRecord s;
        s.specular_ray = Ray{rec.position, rand_unit_vec(MRT_DIM_ISOTROPIC), ray_in.time};
        s.attenuation = albedo->color(rec.u, rec.v, rec.position);
        return s;
    }
};
// VoxelMaterial, hittable/fast_oct_tree/voxel.rs:57-127
struct VoxelMaterial : Material {
    enum class Kind { Lambertian, Reflect, Volume } kind = Kind::Lambertian;
    Rgb color;  // Lambertian color / Reflect albedo / Volume color
    float fuzz = 0.f;
    std::optional<ScatterRecord> scatter(const Ray& ray_in, const HitRay& rec) const override {
        ScatterRecord s;
        switch (kind) {
            case Kind::Lambertian:  // :84-89
                s.attenuation = color;
                s.pdf = PdfKind::Lambertian;
                s.pdf_normal = rec.normal;
                s.scattering_pdf = SpdfKind::Lambertian;
                return s;
            case Kind::Volume:  // :90-95
                s.attenuation = color;
                s.pdf = PdfKind::Isotropic;
                s.scattering_pdf = SpdfKind::VoxelVolume;
                return s;
            default: {  // Reflect, :96-115
                V3 reflected = reflect(normalize(ray_in.direction), rec.normal);
                if (dot(reflected, rec.normal) > 0.0) {
                    s.specular_ray = Ray{rec.position, reflected + (double)fuzz * rand_unit_vec(MRT_DIM_METAL_FUZZ), ray_in.time};
                    s.attenuation = color;
                    return s;
                }
                return std::nullopt;
            }
        }
    }
};

// hittable.rs:414-448 HitRecord::new
inline HitRecord make_hit_record(const Ray& ray, V3 position, V3 normal, double t, double u, double v,
                                 const Material* material) {
    bool front_face = dot(ray.direction, normal) <= 0.0;
    HitRay hr{position, ray.direction, normal, front_face, u, v};
    MaterialEffect eff;
    if (auto e = material->emmit(hr)) {
        eff.kind = EffectKind::Emmit;
        eff.emitted = *e;
    } else if (auto s = material->scatter(ray, hr)) {
        eff.kind = EffectKind::Scatter;
        eff.scatter = *s;
    } else {
        eff.kind = EffectKind::NoEmmit;
    }
    return HitRecord{position, hr.normal, t, front_face, u, v, eff};
}

// ---------------------------------------------------------------- Aabb (bvh.rs:12-71)
struct Aabb {
    V3 minimum, maximum;
    bool hit(const Ray& ray, double t_min, double t_max) const {
        for (int a = 0; a < 3; ++a) {
            double inv_d = 1.0 / ray.direction[a];
            double t0 = (minimum[a] - ray.origin[a]) * inv_d;
            double t1 = (maximum[a] - ray.origin[a]) * inv_d;
            if (inv_d < 0.0) std::swap(t0, t1);
            // the inner `let`s shadow: every axis is tested against the ORIGINAL bounds
            double t_min_a = t0 > t_min ? t0 : t_min;
            double t_max_a = t1 < t_max ? t1 : t_max;
            if (t_max_a < t_min_a) return false;
        }
        return true;
    }
    Aabb surrounding_box(const Aabb& b) const {
        auto pmin = [](double a, double b) { return a < b ? a : b; };
        auto pmax = [](double a, double b) { return a > b ? a : b; };
        return {{pmin(minimum.x, b.minimum.x), pmin(minimum.y, b.minimum.y), pmin(minimum.z, b.minimum.z)},
                {pmax(maximum.x, b.maximum.x), pmax(maximum.y, b.maximum.y), pmax(maximum.z, b.maximum.z)}};
    }
};

struct RayAreaInfo {  // hittable.rs:450-456
    Ray to_area;
    double area;
    V3 direction, normal, end_point;
};

struct Hittable {  // hittable.rs:36-52
    virtual ~Hittable() = default;
    virtual std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max) const = 0;
    virtual std::optional<Aabb> bounding_box(double time_0, double time_1) const = 0;
    virtual bool prob(const Ray& ray, double* out) const = 0;  // false = reference panics (todo!())
    virtual bool generate_ray_in_area(V3 origin, double time, RayAreaInfo* out) const = 0;
};

// ---------------------------------------------------------------- sphere.rs
inline void get_sphere_uv(V3 p, double* u, double* v) {  // sphere.rs:111-115
    double theta = std::acos(-p.y);
    double phi = std::atan2(-p.z, p.x) + M_PI;
    *u = phi / (2.0 * M_PI);
    *v = theta / M_PI;
}
struct Sphere : Hittable {
    double radius;
    V3 origin;
    const Material* material;
    std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max) const override {
        V3 rel_origin = ray.origin - origin;
        double a = dot(ray.direction, ray.direction);
        double half_b = dot(rel_origin, ray.direction);
        double c = dot(rel_origin, rel_origin) - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0) return std::nullopt;
        double sqrt_d = std::sqrt(discriminant);
        double root = (-half_b - sqrt_d) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrt_d) / a;
            if (root < t_min || t_max < root) return std::nullopt;
        }
        V3 position = ray.at(root);
        double u, v;
        get_sphere_uv((position - origin) / radius, &u, &v);
        return make_hit_record(ray, position, (position - origin) / radius, root, u, v, material);
    }
    std::optional<Aabb> bounding_box(double, double) const override {
        return Aabb{origin - V3{radius, radius, radius}, origin + V3{radius, radius, radius}};
    }
    bool prob(const Ray& ray, double* out) const override {  // sphere.rs:63-70
        double area = M_PI * (radius * radius);
        V3 to_light = origin - ray.origin;
        *out = dot(to_light, to_light) / area;
        return true;
    }
    bool generate_ray_in_area(V3 o, double time, RayAreaInfo* out) const override {  // :72-87
        V3 sphere_direction = rand_unit_vec(MRT_DIM_LIGHT_SPHERE);
        V3 end_point = origin + radius * sphere_direction;
        V3 direction = normalize(end_point - o);
        *out = RayAreaInfo{Ray{o, direction, time}, M_PI * (radius * radius), end_point - o,
                           sphere_direction, end_point};
        return true;
    }
};
struct MovingSphere : Hittable {  // sphere.rs:118-196
    V3 center_0, center_1;
    double time_0, time_1, radius;
    const Material* material;
    V3 center(double time) const {
        return center_0 + ((time - time_0) / (time_1 - time_0)) * (center_1 - center_0);
    }
    std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max) const override {
        V3 rel_origin = ray.origin - center(ray.time);
        double a = dot(ray.direction, ray.direction);
        double half_b = dot(rel_origin, ray.direction);
        double c = dot(rel_origin, rel_origin) - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0) return std::nullopt;
        double sqrt_d = std::sqrt(discriminant);
        double root = (-half_b - sqrt_d) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrt_d) / a;
            if (root < t_min || t_max < root) return std::nullopt;
        }
        V3 position = ray.at(root);
        V3 normal = (position - center(ray.time)) / radius;
        double u, v;
        get_sphere_uv(normal, &u, &v);
        return make_hit_record(ray, position, normal, root, u, v, material);
    }
    std::optional<Aabb> bounding_box(double t0, double t1) const override {
        V3 r{radius, radius, radius};
        return Aabb{center(t0) - r, center(t0) + r}.surrounding_box(Aabb{center(t1) - r, center(t1) + r});
    }
    bool prob(const Ray&, double*) const override { return false; }                  // todo!()
    bool generate_ray_in_area(V3, double, RayAreaInfo*) const override { return false; }  // todo!()
};

// ---------------------------------------------------------------- rect.rs
// axis = the constant axis: 2 for XYRect, 1 for XZRect, 0 for YZRect.
struct Rect : Hittable {
    int axis;
    double a0, a1, b0, b1, k, normal_flip;
    const Material* material;
    int ia() const { return axis == 0 ? 1 : 0; }             // first in-plane axis
    int ib() const { return axis == 2 ? 1 : 2; }             // second in-plane axis
    V3 unit() const { return {axis == 0 ? 1.0 : 0.0, axis == 1 ? 1.0 : 0.0, axis == 2 ? 1.0 : 0.0}; }
    static V3 set3(int i0, double v0, int i1, double v1, int i2, double v2) {
        double c[3];
        c[i0] = v0;
        c[i1] = v1;
        c[i2] = v2;
        return {c[0], c[1], c[2]};
    }
    std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max) const override {
        double t = (k - ray.origin[axis]) / ray.direction[axis];
        if (t < t_min || t > t_max) return std::nullopt;
        double a = ray.origin[ia()] + t * ray.direction[ia()];
        double b = ray.origin[ib()] + t * ray.direction[ib()];
        if (a < a0 || a > a1 || b < b0 || b > b1) return std::nullopt;
        double u = (a - a0) / (a1 - a0), v = (b - b0) / (b1 - b0);
        return make_hit_record(ray, ray.at(t), normal_flip * unit(), t, u, v, material);
    }
    std::optional<Aabb> bounding_box(double, double) const override {
        return Aabb{set3(ia(), a0, ib(), b0, axis, k - 0.001), set3(ia(), a1, ib(), b1, axis, k + 0.001)};
    }
    double area() const { return (a1 - a0) * (b1 - b0); }
    bool prob(const Ray& ray, double* out) const override {  // rect.rs:94-106,214-227,339-351
        V3 center = 0.5 * (set3(ia(), a0, ib(), b0, axis, k) + set3(ia(), a1, ib(), b1, axis, k));
        V3 to_light = center - ray.origin;
        double cos_alpha = std::fabs(normalize(to_light)[axis]);
        if (cos_alpha < 0.00001) {
            *out = 0.0;
            return true;
        }
        double distance_squared = dot(to_light, to_light);
        *out = distance_squared / (cos_alpha * area());
        return true;
    }
    bool generate_ray_in_area(V3 o, double time, RayAreaInfo* out) const override {
        V3 end_point = set3(ia(), rand_scalar(a0, a1, MRT_DIM_LIGHT_A), ib(),
                            rand_scalar(b0, b1, MRT_DIM_LIGHT_B), axis, k);
        V3 direction = normalize(end_point - o);
        *out = RayAreaInfo{Ray{o, direction, time}, area(), end_point - o, normal_flip * unit(), end_point};
        return true;
    }
};

// ---------------------------------------------------------------- render_box.rs
struct RenderBox : Hittable {
    V3 box_min, box_max;
    Rect xyp, xym, xzp, xzm, yzp, yzm;
    RenderBox(V3 mn, V3 mx, const Material* m) : box_min(mn), box_max(mx) {  // :21-93
        auto mk = [&](int axis, double a0, double a1, double b0, double b1, double k, bool flip) {
            Rect r;
            r.axis = axis;
            r.a0 = a0, r.a1 = a1, r.b0 = b0, r.b1 = b1, r.k = k;
            r.normal_flip = flip ? -1.0 : 1.0;
            r.material = m;
            return r;
        };
        xyp = mk(2, mn.x, mx.x, mn.y, mx.y, mx.z, false);
        xym = mk(2, mn.x, mx.x, mn.y, mx.y, mn.z, true);
        xzp = mk(1, mn.x, mx.x, mn.z, mx.z, mx.y, false);
        xzm = mk(1, mn.x, mx.x, mn.z, mx.z, mn.y, true);
        yzp = mk(0, mn.y, mx.y, mn.z, mx.z, mx.x, false);
        yzm = mk(0, mn.y, mx.y, mn.z, mx.z, mn.x, true);
    }
    static void check_hit(std::optional<HitRecord>* closest, std::optional<HitRecord> rec) {  // :96-107
        if (rec) {
            if (closest->has_value()) {
                if (rec->t < (*closest)->t) *closest = rec;
            } else {
                *closest = rec;
            }
        }
    }
    std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max) const override {
        std::optional<HitRecord> closest;
        check_hit(&closest, xyp.hit(ray, t_min, t_max));
        check_hit(&closest, xym.hit(ray, t_min, t_max));
        check_hit(&closest, xzp.hit(ray, t_min, t_max));
        check_hit(&closest, xzm.hit(ray, t_min, t_max));
        check_hit(&closest, yzp.hit(ray, t_min, t_max));
        check_hit(&closest, yzm.hit(ray, t_min, t_max));
        return closest;
    }
    std::optional<Aabb> bounding_box(double, double) const override { return Aabb{box_min, box_max}; }
    bool prob(const Ray& ray, double* out) const override {  // :137-155
        double area = 0.0, p;
        (ray.direction.x >= 0.0 ? yzm : yzp).prob(ray, &p);
        area += p;
        (ray.direction.y >= 0.0 ? xzm : xzp).prob(ray, &p);
        area += p;
        (ray.direction.z >= 0.0 ? xym : xyp).prob(ray, &p);
        area += p;
        *out = area;
        return true;
    }
    bool generate_ray_in_area(V3 o, double time, RayAreaInfo* out) const override {  // :157-181
        uint32_t face = rand_u32(0, 3, MRT_DIM_LIGHT_FACE);
        V3 c{(box_min.x + box_max.x) / 2.0, (box_min.y + box_max.y) / 2.0, (box_min.z + box_max.z) / 2.0};
        V3 to_self = c - o;
        if (face == 0) return (to_self.x >= 0.0 ? yzm : yzp).generate_ray_in_area(o, time, out);
        if (face == 1) return (to_self.y >= 0.0 ? xzm : xzp).generate_ray_in_area(o, time, out);
        return (to_self.z >= 0.0 ? xym : xyp).generate_ray_in_area(o, time, out);
    }
};

// ---------------------------------------------------------------- constant_medium.rs
struct ConstantMedium : Hittable {
    const Hittable* boundary;
    const Material* phase_function;
    double neg_inv_density;
    std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max) const override {
        uint32_t ordinal = g_rng.medium_ordinal++;
        uint32_t dim = (g_rng.in_light_query ? MRT_DIM_LIGHT_QUERY_MEDIUM : MRT_DIM_MEDIUM) + ordinal;
        auto hit1 = boundary->hit(ray, -10000000000.0, 10000000000.0);
        if (!hit1) return std::nullopt;
        auto hit2 = boundary->hit(ray, hit1->t + 0.0001, 10000000000.0);
        if (!hit2) return std::nullopt;
        if (hit1->t < t_min) hit1->t = t_min;
        if (hit2->t > t_max) hit2->t = t_max;
        if (hit1->t >= hit2->t) return std::nullopt;
        if (hit1->t < 0.0) hit1->t = 0.0;
        V3 d = ray.direction;
        double ray_length = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
        double distance_inside_boundary = (hit2->t - hit1->t) * ray_length;
        double hit_distance = neg_inv_density * std::log(rand_scalar(0.0, 1.0, dim));
        if (hit_distance > distance_inside_boundary) return std::nullopt;
        double t = hit1->t + hit_distance / ray_length;
        V3 position = ray.at(t);
        HitRay hr{position, ray.direction, V3{1.0, 0.0, 0.0}, true, 0.0, 0.0};
        MaterialEffect eff;
        if (auto e = phase_function->emmit(hr)) {
            eff.kind = EffectKind::Emmit;
            eff.emitted = *e;
        } else if (auto s = phase_function->scatter(ray, hr)) {
            eff.kind = EffectKind::Scatter;
            eff.scatter = *s;
        } else {
            eff.kind = EffectKind::NoEmmit;
        }
        return HitRecord{position, V3{1.0, 0.0, 0.0}, t, false, 0.0, 0.0, eff};
    }
    std::optional<Aabb> bounding_box(double t0, double t1) const override {
        return boundary->bounding_box(t0, t1);
    }
    bool prob(const Ray& ray, double* out) const override { return boundary->prob(ray, out); }
    bool generate_ray_in_area(V3 o, double time, RayAreaInfo* out) const override {
        return boundary->generate_ray_in_area(o, time, out);
    }
};

// ---------------------------------------------------------------- fast_oct_tree/hittable.rs:9-86
struct OctreeShape : Hittable {
    octref::Tree tree;
    std::vector<mrt_voxel> voxels;
    std::vector<VoxelMaterial> voxel_materials;  // the voxel's own material (volume_material() for volumes)
    std::vector<VoxelMaterial> edge_materials;   // solid_material.to_material() of a volume's edge effect
    static double rand01(void*, uint32_t index) { return rand_scalar(0.0, 1.0, MRT_DIM_OCT_VOLUME + index); }
    std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max) const override {
        double ws = (double)tree.world_size();
        Aabb aabb{{0.0, 0.0, 0.0}, {ws, ws, ws}};
        if (!aabb.hit(ray, t_min, t_max)) return std::nullopt;
        octref::Ray r{{ray.origin.x, ray.origin.y, ray.origin.z},
                      {0, 0, 0},
                      ray.time};
        V3 nd = normalize(ray.direction);
        r.direction[0] = nd.x, r.direction[1] = nd.y, r.direction[2] = nd.z;
        octref::HitInfo info;
        octref::VolumeCtx ctx;
        ctx.voxels = voxels.data(), ctx.rand01 = &OctreeShape::rand01, ctx.entry = &g_rng.oct_ordinal;
        if (!tree.trace_ray(r, ctx, &info)) return std::nullopt;
        V3 hp{info.hit_position[0], info.hit_position[1], info.hit_position[2]};
        V3 dp = hp - ray.origin;
        double t = std::sqrt(dot(dp, dp) / dot(ray.direction, ray.direction));
        if (!(t > t_min && t < t_max)) return std::nullopt;
        if (!info.solid)  // volume: normal unit_x, uv origin (hittable.rs:41-60)
            return make_hit_record(ray, hp, V3{1.0, 0.0, 0.0}, t, 0.0, 0.0, &voxel_materials[(size_t)info.voxel]);
        const Material* m = info.edge ? &edge_materials[(size_t)info.voxel] : &voxel_materials[(size_t)info.voxel];
        // solid voxels: axis normal, uv (0.5, 0.5)
        V3 n{info.normal[0], info.normal[1], info.normal[2]};
        return make_hit_record(ray, hp, n, t, 0.5, 0.5, m);
    }
    std::optional<Aabb> bounding_box(double, double) const override {
        if (tree.is_empty()) return std::nullopt;
        double ws = (double)tree.world_size();
        return Aabb{{0.0, 0.0, 0.0}, {ws, ws, ws}};
    }
    bool prob(const Ray&, double*) const override { return false; }                        // todo!()
    bool generate_ray_in_area(V3, double, RayAreaInfo*) const override { return false; }  // todo!()
};

// ---------------------------------------------------------------- Transform + Object (hittable.rs:64-322)
struct Transform {
    M4 world_transform;
    Transform get_inverse() const {
        Transform t;
        if (!invert4(world_transform, &t.world_transform)) std::abort();  // expect("not invertible")
        return t;
    }
    Ray mul_ray(const Ray& ray) const {  // :92-104
        V3 direction_world = ray.origin + ray.direction;
        V4 direction_end = mul(world_transform, V4{direction_world.x, direction_world.y, direction_world.z, 1.0});
        V4 world_origin = mul(world_transform, V4{ray.origin.x, ray.origin.y, ray.origin.z, 1.0});
        V3 direction{direction_end.x - world_origin.x, direction_end.y - world_origin.y,
                     direction_end.z - world_origin.z};
        return Ray{from_homogeneous(world_origin), direction, ray.time};
    }
    static V3 from_homogeneous(V4 v) {
        double s = 1.0 / v.w;
        return {v.x * s, v.y * s, v.z * s};
    }
    V3 mul_point3(V3 p) const { return from_homogeneous(mul(world_transform, V4{p.x, p.y, p.z, 1.0})); }
};
struct Object : Hittable {
    const Hittable* shape;
    Transform transform;
    std::optional<HitRecord> hit(const Ray& ray, double t_min, double t_max) const override {  // :204-237
        Ray shape_ray = transform.mul_ray(ray);
        auto h = shape->hit(shape_ray, t_min, t_max);
        if (!h) return std::nullopt;
        const M4& m = transform.world_transform;
        M3 three{{{m.c[0].x, m.c[0].y, m.c[0].z}, {m.c[1].x, m.c[1].y, m.c[1].z}, {m.c[2].x, m.c[2].y, m.c[2].z}}};
        M3 three_inv;
        if (!invert3(three, &three_inv)) std::abort();
        Transform inv = transform.get_inverse();
        V3 world_position = inv.mul_point3(h->position);
        V3 normal_world = mul(three_inv, h->normal);
        bool front_face = dot(ray.direction, normal_world) <= 0.0;
        return HitRecord{world_position, normal_world, h->t, front_face, h->u, h->v, h->material_effect};
    }
    std::optional<Aabb> bounding_box(double t0, double t1) const override {  // :239-282
        auto aabb = shape->bounding_box(t0, t1);
        if (!aabb) return std::nullopt;
        Transform inv = transform.get_inverse();
        V3 a = inv.mul_point3(aabb->minimum), b = inv.mul_point3(aabb->maximum);
        return Aabb{{a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z},
                    {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}};
    }
    bool prob(const Ray& ray, double* out) const override {  // :284-287 (applies the INVERSE to the ray)
        Transform inv = transform.get_inverse();
        return shape->prob(inv.mul_ray(ray), out);
    }
    bool generate_ray_in_area(V3 origin, double time, RayAreaInfo* out) const override {  // :289-312
        RayAreaInfo o;
        if (!shape->generate_ray_in_area(transform.mul_point3(origin), time, &o)) return false;
        Transform inv = transform.get_inverse();
        V3 end_point = inv.mul_point3(o.end_point);
        Ray to_area = inv.mul_ray(o.to_area);
        V3 scaling_v = inv.mul_point3(V3{1.0, 0.0, 0.0}) - inv.mul_point3(V3{0.0, 0.0, 0.0});
        double scaling = std::fabs(magnitude(scaling_v));
        double area = o.area * scaling;
        V3 direction_end = inv.mul_point3(o.to_area.origin + o.direction);
        V3 direction = direction_end - to_area.origin;
        V3 normal_end = inv.mul_point3(o.to_area.origin + o.normal);
        V3 normal = normal_end - to_area.origin;
        *out = RayAreaInfo{to_area, area, direction, normal, end_point};
        return true;
    }
};

// ---------------------------------------------------------------- BvhTree (bvh.rs:72-294)
struct BvhTreeNode {
    enum Kind { None, Child, Leaf } kind = None;
    Aabb bounding_box_{};
    std::unique_ptr<BvhTreeNode> left, right;
    size_t idx = 0;

    std::optional<Aabb> bounding_box(const std::vector<Object>& objects, double t0, double t1) const {  // :168-182
        switch (kind) {
            case None: return Aabb{{0, 0, 0}, {0, 0, 0}};
            case Child: return bounding_box_;
            default: return objects[idx].bounding_box(t0, t1);
        }
    }
    std::optional<HitRecord> hit(const std::vector<Object>& objects, const Ray& ray, double t_min,
                                 double t_max) const {  // :135-167
        auto bb = bounding_box(objects, t_min, t_max);
        if (!bb) std::abort();  // expect("object does not have bounding box")
        if (!bb->hit(ray, t_min, t_max)) return std::nullopt;
        switch (kind) {
            case None: return std::nullopt;
            case Child: {
                auto left_hit = left->hit(objects, ray, t_min, t_max);
                if (left_hit) {
                    auto right_hit = right->hit(objects, ray, t_min, left_hit->t);
                    if (right_hit) return right_hit;
                    return left_hit;
                }
                return right->hit(objects, ray, t_min, t_max);
            }
            default: {
                auto h = objects[idx].hit(ray, t_min, t_max);
                if (h) h->object = (int)idx;
                return h;
            }
        }
    }
};
struct BuildRng {  // stands in for the reference's global RNG during BvhTreeNode::new (bvh.rs:215)
    uint64_t state;
    uint32_t next_u32() {
        state += 0x9E3779B97F4A7C15ULL;
        return (uint32_t)(mix64(state) >> 32);
    }
};
// bvh.rs:206-293. `objects` is the (sorted) slice of object indices the reference
// passes down; leaves index the ORIGINAL order through start + offset.
std::unique_ptr<BvhTreeNode> bvh_new(const std::vector<size_t>& objects, const std::vector<Object>& full,
                                     size_t start, size_t end, size_t offset, double t0, double t1,
                                     BuildRng* rng) {
    uint32_t axis = rng->next_u32() % 2;  // rand_u32(0, 2)
    size_t span = end - start;
    auto less = [&](size_t a, size_t b) {  // box_compare: Less iff a.min[axis] < b.min[axis]
        auto ab = full[a].bounding_box(0.0, 0.0), bb = full[b].bounding_box(0.0, 0.0);
        if (!ab || !bb) std::abort();
        return ab->minimum[(int)axis] < bb->minimum[(int)axis];
    };
    auto leaf = [](size_t idx) {
        auto n = std::make_unique<BvhTreeNode>();
        n->kind = BvhTreeNode::Leaf;
        n->idx = idx;
        return n;
    };
    std::unique_ptr<BvhTreeNode> left, right;
    if (span == 1) {
        left = leaf(start + offset);
        right = leaf(start + offset);
    } else if (span == 2) {
        if (less(objects[start], objects[start + 1])) {
            left = leaf(start + offset);
            right = leaf(start + 1 + offset);
        } else {
            left = leaf(start + 1 + offset);
            right = leaf(start + offset);
        }
    } else {
        std::vector<size_t> s_vec(objects.begin() + (long)start, objects.begin() + (long)end);
        std::stable_sort(s_vec.begin(), s_vec.end(), less);
        size_t middle = s_vec.size() / 2;
        std::vector<size_t> lo(s_vec.begin(), s_vec.begin() + (long)middle);
        std::vector<size_t> hi(s_vec.begin() + (long)middle, s_vec.end());
        left = bvh_new(lo, full, 0, middle, offset, t0, t1, rng);
        right = bvh_new(hi, full, 0, hi.size(), offset + middle, t0, t1, rng);
    }
    auto lb = left->bounding_box(full, t0, t1), rb = right->bounding_box(full, t0, t1);
    if (!lb || !rb) std::abort();
    auto n = std::make_unique<BvhTreeNode>();
    n->kind = BvhTreeNode::Child;
    n->bounding_box_ = lb->surrounding_box(*rb);
    n->left = std::move(left);
    n->right = std::move(right);
    return n;
}

// ---------------------------------------------------------------- camera.rs:44-106
struct Camera {
    V3 origin, lower_left_corner, horizontal, vertical, u, v;
    double lens_radius, start_time, end_time;
    explicit Camera(const mrt_camera_info& info) {
        double theta = info.fov * M_PI / 180.0;
        double h = std::tan(theta / 2.0);
        double world_height = 2.0 * h;
        double world_width = info.aspect_ratio * world_height;
        V3 o{info.origin[0], info.origin[1], info.origin[2]};
        V3 look_at{info.look_at[0], info.look_at[1], info.look_at[2]};
        V3 up{info.up_vector[0], info.up_vector[1], info.up_vector[2]};
        V3 w = normalize(o - look_at);
        u = normalize(cross(up, w));
        v = cross(w, u);
        horizontal = info.focus_distance * world_width * u;
        vertical = info.focus_distance * world_height * v;
        origin = o;
        lower_left_corner = o - horizontal / 2.0 - vertical / 2.0 - info.focus_distance * w;
        lens_radius = info.aperture / 2.0;
        start_time = info.start_time;
        end_time = info.end_time;
    }
    static V3 random_in_unit_disk() {
        for (uint32_t k = 0; k < MRT_MAX_REJECT; ++k) {
            V3 p{rand_scalar(-1.0, 1.0, MRT_DIM_LENS + 2 * k), rand_scalar(-1.0, 1.0, MRT_DIM_LENS + 2 * k + 1), 0.0};
            if (dot(p, p) < 1.0) return p;
        }
        return {0.0, 0.0, 0.0};
    }
    Ray get_ray(double s, double t) const {
        V3 rd = lens_radius * random_in_unit_disk();
        V3 offset = u * rd.x + v * rd.y;
        return Ray{origin, lower_left_corner + s * horizontal + t * vertical - origin - offset,
                   rand_scalar(start_time, end_time, MRT_DIM_TIME)};
    }
};

// ---------------------------------------------------------------- backgrounds
struct Background {
    mrt_background b;
    Rgb sky(const Ray& ray) const {  // sky.rs:13-28
        V3 unit = normalize(ray.direction);
        double t = 0.5 * (unit.y + 1.0);
        Rgb color = mul_f64(Rgb{1.0f, 1.0f, 1.0f}, 1.0 - t) + mul_f64(Rgb{0.5f, 0.7f, 1.0f}, t);
        return mul_f64(color, b.intensity);
    }
    Rgb color(const Ray& ray) const {
        switch (b.kind) {
            case MRT_BG_SKY: return sky(ray);
            case MRT_BG_SUN_SKY: {  // sun_sky.rs:32-60
                double r = std::cos(b.sun_phi);
                V3 sun_ray{r * std::cos(b.sun_theta), std::sin(b.sun_phi), r * std::sin(b.sun_theta)};
                double sun_cos = dot(sun_ray, normalize(ray.direction));
                if (sun_cos > std::cos(b.sun_radius) && sun_cos > 0.0)
                    return mul_f64(Rgb{1.0f, 1.0f, 1.0f}, b.sun_brightness);
                return sky(ray);
            }
            default: return Rgb{b.color[0], b.color[1], b.color[2]};  // constant_color.rs:12-14
        }
    }
};

// ---------------------------------------------------------------- World (world.rs:79-114)
struct World {
    std::vector<mrt_perlin> perlins;
    std::vector<std::unique_ptr<Image>> images;
    std::vector<std::unique_ptr<Texture>> textures;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Hittable>> shapes;
    std::vector<Object> objects;
    std::unique_ptr<BvhTreeNode> root;
    std::vector<Object> lights;
    Background background;
    std::unique_ptr<Camera> camera;
    mrt_sun sun;

    std::optional<HitRecord> nearest_hit(const Ray& ray, float t_min, float t_max) const {  // :112-114
        g_rng.rays++;
        g_rng.medium_ordinal = 0;
        g_rng.oct_ordinal = 0;
        g_rng.in_light_query = false;
        if (!g_walk_steps) return root->hit(objects, ray, (double)t_min, (double)t_max);
        const unsigned long long before = octref::oct_step_counter();
        auto rec = root->hit(objects, ray, (double)t_min, (double)t_max);
        g_walk_steps->push_back((uint32_t)(octref::oct_step_counter() - before));
        return rec;
    }
    // :98-110; returns the index of the light instead of a clone of it
    std::optional<std::pair<size_t, HitRecord>> nearest_light_hit(const Ray& ray, double t_min, double t_max) const {
        g_rng.light_queries++;
        g_rng.medium_ordinal = 0;
        g_rng.in_light_query = true;
        std::optional<std::pair<size_t, HitRecord>> acc;
        for (size_t i = 0; i < lights.size(); ++i) {
            auto h = lights[i].hit(ray, t_min, t_max);
            if (!h) continue;
            if (!acc)
                acc = std::make_pair(i, *h);
            else if (!(acc->second.t < h->t))
                acc = std::make_pair(i, *h);
        }
        g_rng.in_light_query = false;
        return acc;
    }
};

// ---------------------------------------------------------------- pdf.rs
struct PdfSample {
    V3 direction;
    double value;
};
inline std::optional<double> cosine_pdf_value(const OrthoNormalBasis& uvw, const Ray& ray) {  // :32-39
    double c = dot(ray.direction, uvw.w());
    return c <= 0.0 ? 0.0 : c / M_PI;
}
inline std::optional<PdfSample> cosine_pdf_generate(const OrthoNormalBasis& uvw) {  // :45-60
    V3 direction = normalize(uvw.local(random_cosine_direction()));
    double c = dot(direction, uvw.w());
    double value = c <= 0.0 ? 0.0 : c / M_PI;
    return PdfSample{direction, value};
}
inline std::optional<double> light_pdf_value(const Ray& ray, const World& world) {  // :82-99
    auto lh = world.nearest_light_hit(ray, ray.time, std::numeric_limits<double>::max());
    if (!lh) return std::nullopt;
    const HitRecord& hit = lh->second;
    V3 to_light = hit.position - ray.origin;
    double light_cos = std::fabs(dot(normalize(to_light), hit.normal));
    if (light_cos >= 0.000001) {
        double p;
        if (!world.lights[lh->first].prob(Ray{ray.origin, normalize(to_light), hit.t}, &p)) std::abort();
        return p;
    }
    return 0.0;
}
inline std::optional<PdfSample> light_pdf_generate(const Ray& incoming, V3 hit_point, const World& world) {  // :104-130
    if (world.lights.empty()) return std::nullopt;
    size_t idx = rand_u32(0, (uint32_t)world.lights.size(), MRT_DIM_LIGHT_INDEX);
    RayAreaInfo ai;
    if (!world.lights[idx].generate_ray_in_area(hit_point, incoming.time, &ai)) std::abort();
    double light_cos = std::fabs(dot(normalize(ai.to_area.direction), ai.normal));
    double dist_squared = dot(ai.direction, ai.direction);
    if (light_cos >= 0.000001) {
        double value = dist_squared / (light_cos * ai.area);
        return PdfSample{normalize(ai.to_area.direction), value};
    }
    return std::nullopt;
}
inline V3 sun_direction(const mrt_sun& s) {  // sun.rs:17-20
    double r = std::cos(s.phi);
    return {r * std::cos(s.theta), std::sin(s.phi), r * std::sin(s.theta)};
}
inline std::optional<double> sky_pdf_value(const Ray& direction, const World& world) {  // :135-148
    if (!world.sun.has_sun) return 1.0;
    V3 sv = sun_direction(world.sun);
    return dot(direction.direction, sv) >= std::cos(world.sun.radius) ? 1.0 : 0.0;
}
inline std::optional<PdfSample> sky_pdf_generate(const World& world) {  // :154-193
    if (!world.sun.has_sun) return PdfSample{rand_unit_vec(MRT_DIM_SKY), 4.0 * M_PI};
    double rand_r = rand_scalar(0.0, 1.0, MRT_DIM_SUN_R);
    double rand_theta = rand_scalar(0.0, 2.0 * M_PI, MRT_DIM_SUN_THETA);
    double r = std::sqrt(rand_r), theta = rand_theta;
    r = r * world.sun.radius;
    V3 sv = sun_direction(world.sun);
    V3 cross_vector = cross(sv, V3{0.0, 0.0, 1.0});
    if (magnitude(cross_vector) < 0.01) std::abort();  // panic!()
    V3 k = normalize(cross_vector);
    V3 v_rot = std::cos(r) * sv + cross(k, sv) * std::sin(theta) + k * dot(k, sv) * (1.0 - std::cos(theta));
    double area = M_PI * (world.sun.radius * world.sun.radius);
    return PdfSample{v_rot, area / (4.0 * M_PI)};
}
// LambertianPDF::generate, pdf.rs:234-282
inline std::optional<PdfSample> lambertian_pdf_generate(V3 normal, const Ray& incoming, V3 hit_point,
                                                        const World& world) {
    OrthoNormalBasis uvw = OrthoNormalBasis::build_from_w(normal);
    uint32_t r = rand_u32(0, 3, MRT_DIM_PDF_CHOICE);
    std::optional<PdfSample> v;
    if (r == 0)
        v = cosine_pdf_generate(uvw);
    else if (r == 1)
        v = light_pdf_generate(incoming, hit_point, world);
    else
        v = sky_pdf_generate(world);
    if (!v) return std::nullopt;
    Ray value_ray{hit_point, v->direction, 0.0};
    std::optional<double> values[2];
    if (r == 0) {
        values[0] = light_pdf_value(value_ray, world);
        values[1] = sky_pdf_value(value_ray, world);
    } else if (r == 1) {
        values[0] = cosine_pdf_value(uvw, value_ray);
        values[1] = sky_pdf_value(value_ray, world);
    } else {
        values[0] = light_pdf_value(value_ray, world);
        values[1] = cosine_pdf_value(uvw, value_ray);
    }
    double sum = 0.0;
    int total = 0;
    for (auto& x : values)
        if (x) {
            sum += *x;
            total += 1;
        }
    if (total != 0) return PdfSample{v->direction, (sum + v->value) / (double)(total + 1)};
    return std::nullopt;
}

// ---------------------------------------------------------------- RayTracingShader::ray_color (ray_tracer.rs:159-303)
Rgb ray_color(const Ray& ray, const World& world, uint32_t depth) {
    if (depth == 0) return BLACK;
    g_rng.seg = 50 - depth;
    auto rec = world.nearest_hit(ray, 0.001f, std::numeric_limits<float>::max());
    if (!rec) return world.background.color(ray);
    const MaterialEffect& eff = rec->material_effect;
    switch (eff.kind) {
        case EffectKind::Emmit: return eff.emitted;
        case EffectKind::NoEmmit: return BLACK;
        default: break;
    }
    const ScatterRecord& s = eff.scatter;
    if (s.specular_ray) {
        Rgb c = ray_color(*s.specular_ray, world, depth - 1);
        return s.attenuation * c;
    }
    std::optional<PdfSample> g;
    if (s.pdf == PdfKind::Lambertian)
        g = lambertian_pdf_generate(s.pdf_normal, ray, rec->position, world);
    else if (s.pdf == PdfKind::Isotropic)
        g = PdfSample{rand_unit_vec(MRT_DIM_ISOTROPIC_PDF), 1.0};  // pdf.rs:71-78
    else
        std::abort();  // expect("if material is not specular there should be a pdf")
    if (!g) return BLACK;
    Ray scattered{rec->position, g->direction, rec->t};
    // Lambertian::scattering_pdf_fn, material.rs:30-41
    double cosine = dot(rec->normal, normalize(scattered.direction));
    if (cosine < 0.0) return BLACK;
    double scattering_pdf = cosine / M_PI;
    if (scattering_pdf == 0.0) return BLACK;
    double value = g->value / scattering_pdf;
    Rgb c = ray_color(scattered, world, depth - 1);
    return div_f64(s.attenuation * c, value);
}

// ---------------------------------------------------------------- DiffuseShader::ray_color (ray_tracer.rs:126-157)
Rgb diffuse_color(const Ray& ray, const World& world) {
    g_rng.seg = 0;
    auto rec = world.nearest_hit(ray, 0.001f, std::numeric_limits<float>::max());
    if (!rec) return world.background.color(ray);
    const MaterialEffect& eff = rec->material_effect;
    switch (eff.kind) {
        case EffectKind::Emmit: return eff.emitted;
        case EffectKind::NoEmmit: return BLACK;
        default: return eff.scatter.attenuation;
    }
}

// ---------------------------------------------------------------- LightMapShader::ray_color (ray_tracer.rs:83-124)
// Random numbers: the camera ray and the first query are segment 0; the sample on light i and the
// query towards it are segment 1 + i (the reference draws fresh numbers for every light).
Rgb light_map_color(const Ray& ray, const World& world) {
    g_rng.seg = 0;
    auto record = world.nearest_hit(ray, 0.001f, std::numeric_limits<float>::max());
    if (!record) return BLACK;
    Rgb acc = BLACK;
    for (size_t i = 0; i < world.lights.size(); ++i) {
        g_rng.seg = 1 + (uint32_t)i;
        RayAreaInfo area;
        if (!world.lights[i].generate_ray_in_area(record->position, record->t, &area)) std::abort();
        Rgb x = BLACK;
        if (auto r = world.nearest_hit(area.to_area, 0.001f, std::numeric_limits<float>::max())) {
            V3 t = area.end_point - r->position;
            double m = std::sqrt(dot(t, t));
            Rgb o = mul_f64(Rgb{1.f, 1.f, 1.f}, m);
            x = (o.r != o.r || o.g != o.g || o.b != o.b) ? Rgb{1.f, 0.f, 0.f} : o;
        }
        acc = acc + x;
    }
    return acc;
}

struct Scene {
    World world;
    uint64_t bvh_seed;
    int shader = MRT_SHADER_RAYTRACING;  // CurrentShader, ray_tracer.rs:304-309
};

bool fail(char* err, size_t errlen, const std::string& msg) {
    if (err && errlen) {
        std::snprintf(err, errlen, "%s", msg.c_str());
    }
    return false;
}

bool build_scene(const mrt_scene_desc* d, Scene* s, char* err, size_t errlen) {
    World& w = s->world;
    for (uint32_t i = 0; i < d->n_images; ++i) {
        auto im = std::make_unique<Image>();
        im->width = d->images[i].width;
        im->height = d->images[i].height;
        if (im->width <= 0 || im->height <= 0 || !d->images[i].rgb) return fail(err, errlen, "bad image");
        im->rgb.assign(d->images[i].rgb, d->images[i].rgb + (size_t)im->width * im->height * 3);
        w.images.push_back(std::move(im));
    }
    if (d->n_perlins && !d->perlins) return fail(err, errlen, "perlins is NULL");
    w.perlins.assign(d->perlins, d->perlins + d->n_perlins);
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const mrt_texture& t = d->textures[i];
        auto child = [&](int32_t c) -> const Texture* {
            return (c >= 0 && (uint32_t)c < i) ? w.textures[(size_t)c].get() : nullptr;
        };
        switch (t.kind) {
            case MRT_TEX_SOLID: {
                auto x = std::make_unique<SolidColor>();
                x->c = {t.color[0], t.color[1], t.color[2]};
                w.textures.push_back(std::move(x));
                break;
            }
            case MRT_TEX_IMAGE: {
                if (t.image < 0 || (uint32_t)t.image >= d->n_images) return fail(err, errlen, "bad image index");
                auto x = std::make_unique<ImageTexture>();
                x->img = w.images[(size_t)t.image].get();
                w.textures.push_back(std::move(x));
                break;
            }
            case MRT_TEX_MULTIPLY: {
                auto x = std::make_unique<MultiplyTexture>();
                x->a = child(t.a), x->b = child(t.b);
                if (!x->a || !x->b) return fail(err, errlen, "bad texture child");
                w.textures.push_back(std::move(x));
                break;
            }
            case MRT_TEX_CHECKER: {
                auto x = std::make_unique<CheckerTexture>();
                x->odd = child(t.a), x->even = child(t.b);
                if (!x->odd || !x->even) return fail(err, errlen, "bad texture child");
                w.textures.push_back(std::move(x));
                break;
            }
            case MRT_TEX_DEBUG_V: w.textures.push_back(std::make_unique<DebugV>()); break;
            case MRT_TEX_PERLIN: {
                if (t.image < 0 || (uint32_t)t.image >= d->n_perlins) return fail(err, errlen, "bad perlin index");
                auto x = std::make_unique<PerlinTexture>();
                x->pn = &w.perlins[(size_t)t.image];
                w.textures.push_back(std::move(x));
                break;
            }
            default: return fail(err, errlen, "unknown texture kind");
        }
    }
    auto tex = [&](int32_t i) -> const Texture* {
        return (i >= 0 && (uint32_t)i < d->n_textures) ? w.textures[(size_t)i].get() : nullptr;
    };
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const mrt_material& m = d->materials[i];
        switch (m.kind) {
            case MRT_MAT_LAMBERTIAN: {
                auto x = std::make_unique<Lambertian>();
                if (!(x->albedo = tex(m.texture))) return fail(err, errlen, "bad material texture");
                w.materials.push_back(std::move(x));
                break;
            }
            case MRT_MAT_METAL: {
                auto x = std::make_unique<Metal>();
                if (!(x->albedo = tex(m.texture))) return fail(err, errlen, "bad material texture");
                x->fuzz = m.fuzz;
                w.materials.push_back(std::move(x));
                break;
            }
            case MRT_MAT_DIELECTRIC: {
                auto x = std::make_unique<Dielectric>();
                x->index_refraction = m.index_refraction;
                x->color = {m.color[0], m.color[1], m.color[2]};
                w.materials.push_back(std::move(x));
                break;
            }
            case MRT_MAT_DIFFUSE_LIGHT: {
                auto x = std::make_unique<DiffuseLight>();
                if (!(x->emit = tex(m.texture))) return fail(err, errlen, "bad material texture");
                w.materials.push_back(std::move(x));
                break;
            }
            case MRT_MAT_ISOTROPIC: {
                auto x = std::make_unique<Isotropic>();
                if (!(x->albedo = tex(m.texture))) return fail(err, errlen, "bad material texture");
                w.materials.push_back(std::move(x));
                break;
            }
            default: return fail(err, errlen, "unknown material kind");
        }
    }
    auto mat = [&](int32_t i) -> const Material* {
        return (i >= 0 && (uint32_t)i < d->n_materials) ? w.materials[(size_t)i].get() : nullptr;
    };
    for (uint32_t i = 0; i < d->n_shapes; ++i) {
        const mrt_shape& sh = d->shapes[i];
        const double* p = sh.p;
        switch (sh.kind) {
            case MRT_SHAPE_SPHERE: {
                auto x = std::make_unique<Sphere>();
                x->radius = p[0];
                x->origin = {p[1], p[2], p[3]};
                if (!(x->material = mat(sh.material))) return fail(err, errlen, "bad shape material");
                w.shapes.push_back(std::move(x));
                break;
            }
            case MRT_SHAPE_MOVING_SPHERE: {
                auto x = std::make_unique<MovingSphere>();
                x->radius = p[0];
                x->center_0 = {p[1], p[2], p[3]};
                x->center_1 = {p[4], p[5], p[6]};
                x->time_0 = p[7];
                x->time_1 = p[8];
                if (!(x->material = mat(sh.material))) return fail(err, errlen, "bad shape material");
                w.shapes.push_back(std::move(x));
                break;
            }
            case MRT_SHAPE_XY_RECT:
            case MRT_SHAPE_XZ_RECT:
            case MRT_SHAPE_YZ_RECT: {
                auto x = std::make_unique<Rect>();
                x->axis = sh.kind == MRT_SHAPE_XY_RECT ? 2 : (sh.kind == MRT_SHAPE_XZ_RECT ? 1 : 0);
                x->a0 = p[0], x->a1 = p[1], x->b0 = p[2], x->b1 = p[3], x->k = p[4], x->normal_flip = p[5];
                if (!(x->material = mat(sh.material))) return fail(err, errlen, "bad shape material");
                w.shapes.push_back(std::move(x));
                break;
            }
            case MRT_SHAPE_RENDER_BOX: {
                const Material* m = mat(sh.material);
                if (!m) return fail(err, errlen, "bad shape material");
                w.shapes.push_back(std::make_unique<RenderBox>(V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}, m));
                break;
            }
            case MRT_SHAPE_CONSTANT_MEDIUM: {
                if (sh.boundary < 0 || (uint32_t)sh.boundary >= i) return fail(err, errlen, "bad medium boundary");
                auto x = std::make_unique<ConstantMedium>();
                x->boundary = w.shapes[(size_t)sh.boundary].get();
                if (!(x->phase_function = mat(sh.material))) return fail(err, errlen, "bad phase material");
                x->neg_inv_density = -1.0 / p[0];
                w.shapes.push_back(std::move(x));
                break;
            }
            case MRT_SHAPE_OCTREE: {
                if (sh.octree < 0 || (uint32_t)sh.octree >= d->n_octrees) return fail(err, errlen, "bad octree index");
                auto x = std::make_unique<OctreeShape>();
                const mrt_octree& o = d->octrees[sh.octree];
                std::string e;
                if (!x->tree.from_linear(o, &e)) return fail(err, errlen, e);
                for (uint32_t v = 0; v < o.n_voxels; ++v) {
                    const mrt_voxel& vx = o.voxels[v];
                    VoxelMaterial own, edge;
                    own.color = {vx.albedo[0], vx.albedo[1], vx.albedo[2]};
                    own.fuzz = vx.fuzz;
                    switch (vx.kind) {
                        case MRT_VOX_LAMBERTIAN: own.kind = VoxelMaterial::Kind::Lambertian; break;
                        case MRT_VOX_REFLECT: own.kind = VoxelMaterial::Kind::Reflect; break;
                        case MRT_VOX_VOLUME: own.kind = VoxelMaterial::Kind::Volume; break;
                        default: return fail(err, errlen, "unknown voxel kind");
                    }
                    edge.color = {vx.edge_albedo[0], vx.edge_albedo[1], vx.edge_albedo[2]};
                    edge.fuzz = vx.edge_fuzz;
                    edge.kind = vx.edge_kind == MRT_VOX_EDGE_REFLECT ? VoxelMaterial::Kind::Reflect : VoxelMaterial::Kind::Lambertian;
                    x->voxels.push_back(vx);
                    x->voxel_materials.push_back(own);
                    x->edge_materials.push_back(edge);
                }
                w.shapes.push_back(std::move(x));
                break;
            }
            default: return fail(err, errlen, "unknown shape kind");
        }
    }
    auto mk_object = [&](const mrt_object& o, Object* out) {
        if (o.shape < 0 || (uint32_t)o.shape >= d->n_shapes) return false;
        out->shape = w.shapes[(size_t)o.shape].get();
        for (int c = 0; c < 4; ++c)
            out->transform.world_transform.c[c] = {o.world_transform[c * 4 + 0], o.world_transform[c * 4 + 1],
                                                   o.world_transform[c * 4 + 2], o.world_transform[c * 4 + 3]};
        M4 tmp;
        return invert4(out->transform.world_transform, &tmp);
    };
    w.objects.resize(d->n_objects);
    for (uint32_t i = 0; i < d->n_objects; ++i)
        if (!mk_object(d->objects[i], &w.objects[i])) return fail(err, errlen, "bad object (shape index or singular transform)");
    w.lights.resize(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; ++i)
        if (!mk_object(d->lights[i], &w.lights[i])) return fail(err, errlen, "bad light");
    w.background.b = d->background;
    w.sun = d->sun;
    w.camera = std::make_unique<Camera>(d->camera);
    // BvhTree::new, bvh.rs:78-96
    if (w.objects.empty()) {
        w.root = std::make_unique<BvhTreeNode>();
    } else {
        for (auto& o : w.objects)
            if (!o.bounding_box(0.0, 0.0)) return fail(err, errlen, "object without bounding box");
        std::vector<size_t> idx(w.objects.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
        BuildRng rng{d->bvh_seed};
        w.root = bvh_new(idx, w.objects, 0, idx.size(), 0, d->camera.start_time, d->camera.end_time, &rng);
    }
    s->bvh_seed = d->bvh_seed;
    return true;
}

void flatten_bvh(const BvhTreeNode* n, const std::vector<Object>& objs, double t0, double t1,
                 std::vector<mrt_bvh_node>* out) {
    size_t me = out->size();
    out->push_back(mrt_bvh_node{});
    mrt_bvh_node rec{};
    if (n->kind == BvhTreeNode::Leaf) {
        auto bb = objs[n->idx].bounding_box(t0, t1);
        rec.object = (int32_t)n->idx;
        rec.bmin[0] = bb->minimum.x, rec.bmin[1] = bb->minimum.y, rec.bmin[2] = bb->minimum.z;
        rec.bmax[0] = bb->maximum.x, rec.bmax[1] = bb->maximum.y, rec.bmax[2] = bb->maximum.z;
    } else if (n->kind == BvhTreeNode::Child) {
        rec.object = -1;
        const Aabb& bb = n->bounding_box_;
        rec.bmin[0] = bb.minimum.x, rec.bmin[1] = bb.minimum.y, rec.bmin[2] = bb.minimum.z;
        rec.bmax[0] = bb.maximum.x, rec.bmax[1] = bb.maximum.y, rec.bmax[2] = bb.maximum.z;
        flatten_bvh(n->left.get(), objs, t0, t1, out);
        flatten_bvh(n->right.get(), objs, t0, t1, out);
    } else {
        rec.object = -1;
    }
    rec.skip = (int32_t)out->size();
    (*out)[me] = rec;
}

// RayTracer::trace_part, ray_tracer.rs:435-463, over columns [x_begin, x_end)
void trace_part(const World& world, int shader, float* accum, uint32_t W, uint32_t H, uint32_t x_begin, uint32_t x_end,
                uint64_t seed, uint64_t sample_index) {
    for (uint32_t x = x_begin; x < x_end; ++x) {
        for (uint32_t y = 0; y < H; ++y) {
            g_rng.key = path_key(seed, (uint64_t)x + (uint64_t)y * W, sample_index);
            g_rng.seg = 0;
            double u = ((double)x + rand_scalar(0.0, 1.0, MRT_DIM_JITTER_U)) / ((double)W - 1.0);
            double v = ((double)y + rand_scalar(0.0, 1.0, MRT_DIM_JITTER_V)) / ((double)H - 1.0);
            Ray r = world.camera->get_ray(u, v);
            Rgb c = shader == MRT_SHADER_DIFFUSE    ? diffuse_color(r, world)      // ray_tracer.rs:448-454
                    : shader == MRT_SHADER_LIGHT_MAP ? light_map_color(r, world)
                                                     : ray_color(r, world, 50);
            float* px = accum + ((size_t)x + (size_t)y * W) * 3;  // parallel_image.rs:328-331
            px[0] += c.r;
            px[1] += c.g;
            px[2] += c.b;
        }
    }
}

}  // namespace

// ================================================================ exported C symbols (checker API)
extern "C" {

void* oracle_scene_create(const mrt_scene_desc* desc, char* err, size_t errlen) {
    if (!desc || desc->abi_version != MRT_ABI_VERSION) {
        fail(err, errlen, "bad descriptor / ABI version");
        return nullptr;
    }
    auto s = std::make_unique<Scene>();
    if (!build_scene(desc, s.get(), err, errlen)) return nullptr;
    return s.release();
}
void oracle_scene_destroy(void* scene) { delete static_cast<Scene*>(scene); }

// RayTracer::set_shader (ray_tracer.rs:409-411): MRT_SHADER_*
int oracle_set_shader(void* scene, int shader) {
    if (!scene || shader < MRT_SHADER_RAYTRACING || shader > MRT_SHADER_LIGHT_MAP) return MRT_ERR_INVALID_ARGUMENT;
    static_cast<Scene*>(scene)->shader = shader;
    return MRT_OK;
}

int oracle_scene_bvh(void* scene, mrt_bvh_node* out, size_t capacity, size_t* n_nodes) {
    auto* s = static_cast<Scene*>(scene);
    std::vector<mrt_bvh_node> nodes;
    flatten_bvh(s->world.root.get(), s->world.objects, s->world.camera->start_time, s->world.camera->end_time, &nodes);
    if (n_nodes) *n_nodes = nodes.size();
    if (out) {
        if (capacity < nodes.size()) return MRT_ERR_INVALID_ARGUMENT;
        std::memcpy(out, nodes.data(), nodes.size() * sizeof(mrt_bvh_node));
    }
    return MRT_OK;
}

int oracle_camera_rays(void* scene, uint32_t width, uint32_t height, const int32_t* pixels, size_t n,
                       uint64_t seed, uint64_t sample_index, double* rays_out) {
    auto* s = static_cast<Scene*>(scene);
    g_rng.mode = 0;
    for (size_t i = 0; i < n; ++i) {
        uint32_t x = (uint32_t)pixels[2 * i], y = (uint32_t)pixels[2 * i + 1];
        g_rng.key = path_key(seed, (uint64_t)x + (uint64_t)y * width, sample_index);
        g_rng.seg = 0;
        double u = ((double)x + rand_scalar(0.0, 1.0, MRT_DIM_JITTER_U)) / ((double)width - 1.0);
        double v = ((double)y + rand_scalar(0.0, 1.0, MRT_DIM_JITTER_V)) / ((double)height - 1.0);
        Ray r = s->world.camera->get_ray(u, v);
        double* o = rays_out + 7 * i;
        o[0] = r.origin.x, o[1] = r.origin.y, o[2] = r.origin.z;
        o[3] = r.direction.x, o[4] = r.direction.y, o[5] = r.direction.z;
        o[6] = r.time;
    }
    return MRT_OK;
}

int oracle_trace_rays(void* scene, const double* rays, size_t n, uint64_t seed, int32_t* obj_id, double* t,
                      double* normal, double* position, int32_t* front_face) {
    auto* s = static_cast<Scene*>(scene);
    const World& w = s->world;
    g_rng.mode = 0;
    for (size_t i = 0; i < n; ++i) {
        const double* r = rays + 7 * i;
        Ray ray{{r[0], r[1], r[2]}, {r[3], r[4], r[5]}, r[6]};
        g_rng.key = path_key(seed, i, 0);
        g_rng.seg = 0;
        auto rec = w.nearest_hit(ray, 0.001f, std::numeric_limits<float>::max());
        int32_t id = rec ? rec->object : -1;
        if (obj_id) obj_id[i] = id;
        if (t) t[i] = rec ? rec->t : 0.0;
        if (normal) {
            normal[3 * i] = rec ? rec->normal.x : 0.0;
            normal[3 * i + 1] = rec ? rec->normal.y : 0.0;
            normal[3 * i + 2] = rec ? rec->normal.z : 0.0;
        }
        if (position) {
            position[3 * i] = rec ? rec->position.x : 0.0;
            position[3 * i + 1] = rec ? rec->position.y : 0.0;
            position[3 * i + 2] = rec ? rec->position.z : 0.0;
        }
        if (front_face) front_face[i] = rec ? (rec->front_face ? 1 : 0) : 0;
    }
    return MRT_OK;
}

// spp x trace_part over the strip, split over `nthreads` vertical sub-strips as
// threaded_render does (ray_tracer.rs:496-533). rng_mode 0 = counter-based
// (shared with the device path), 1 = independent sequential mt19937_64.
int oracle_render_accumulate(void* scene, float* accum_rgb, uint32_t width, uint32_t height, uint32_t x0,
                             uint32_t strip_w, uint32_t spp, uint64_t seed, uint64_t first_sample_index,
                             uint32_t nthreads, int rng_mode, mrt_stats* stats) {
    auto* s = static_cast<Scene*>(scene);
    if (!s || !accum_rgb || width < 2 || height < 2 || x0 + strip_w > width) return MRT_ERR_INVALID_ARGUMENT;
    if (nthreads == 0) nthreads = 1;
    if (nthreads > strip_w) nthreads = strip_w ? strip_w : 1;
    std::vector<std::thread> threads;
    std::atomic<uint64_t> rays{0}, lq{0};
    uint32_t slice = strip_w / nthreads;
    for (uint32_t th = 0; th < nthreads; ++th) {
        uint32_t xb = x0 + th * slice;
        uint32_t xe = (th + 1 == nthreads) ? x0 + strip_w : xb + slice;
        threads.emplace_back([=, &rays, &lq]() {
            g_rng = RngCtx{};
            g_rng.mode = rng_mode;
            g_rng.mt.seed(mix64(seed ^ (0xA5A5A5A5ULL + th)) + first_sample_index);
            for (uint32_t k = 0; k < spp; ++k)
                trace_part(s->world, s->shader, accum_rgb, width, height, xb, xe, seed, first_sample_index + k);
            rays += g_rng.rays;
            lq += g_rng.light_queries;
        });
    }
    for (auto& t : threads) t.join();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->paths = (uint64_t)strip_w * height * spp;
        stats->rays = rays.load();
        stats->light_queries = lq.load();
    }
    return MRT_OK;
}

// Radiance of single pixel-samples (for path-level comparison): out n x 3.
int oracle_sample_radiance(void* scene, uint32_t width, uint32_t height, const int32_t* pixels, size_t n,
                           uint64_t seed, uint64_t sample_index, float* out_rgb) {
    auto* s = static_cast<Scene*>(scene);
    g_rng = RngCtx{};
    for (size_t i = 0; i < n; ++i) {
        uint32_t x = (uint32_t)pixels[2 * i], y = (uint32_t)pixels[2 * i + 1];
        g_rng.key = path_key(seed, (uint64_t)x + (uint64_t)y * width, sample_index);
        g_rng.seg = 0;
        double u = ((double)x + rand_scalar(0.0, 1.0, MRT_DIM_JITTER_U)) / ((double)width - 1.0);
        double v = ((double)y + rand_scalar(0.0, 1.0, MRT_DIM_JITTER_V)) / ((double)height - 1.0);
        Ray r = s->world.camera->get_ray(u, v);
        Rgb c = ray_color(r, s->world, 50);
        out_rgb[3 * i] = c.r, out_rgb[3 * i + 1] = c.g, out_rgb[3 * i + 2] = c.b;
    }
    return MRT_OK;
}

// Steps of the octree walk (ray_iteration, fast_oct_tree/ray_trace.rs:200) that one nearest_hit of each ray takes: a
// diagnostic for walk-length statistics, not a result of the reference.
int oracle_octree_steps(void* scene, const double* rays, size_t n, uint64_t seed, uint32_t* steps_out) {
    auto* s = static_cast<Scene*>(scene);
    if (!s || !rays || !steps_out) return MRT_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; ++i) {
        g_rng = RngCtx{};
        g_rng.key = path_key(seed, i, 0);
        const double* r = rays + 7 * i;
        Ray ray{{r[0], r[1], r[2]}, {r[3], r[4], r[5]}, r[6]};
        const unsigned long long before = octref::oct_step_counter();
        (void)s->world.nearest_hit(ray, 0.001f, std::numeric_limits<float>::max());
        steps_out[i] = (uint32_t)(octref::oct_step_counter() - before);
    }
    return MRT_OK;
}

// Diagnostic: octree walk steps of every segment of the paths of n pixel-samples (the same paths oracle_sample_radiance
// traces): steps_out[i * max_segments + k] = steps of segment k of path i, or 0xFFFFFFFF past the path's end.
int oracle_path_walk_steps(void* scene, uint32_t width, uint32_t height, const int32_t* pixels, size_t n, uint64_t seed,
                           uint64_t sample_index, uint32_t max_segments, uint32_t* steps_out) {
    auto* s = static_cast<Scene*>(scene);
    if (!s || !pixels || !steps_out) return MRT_ERR_INVALID_ARGUMENT;
    std::vector<uint32_t> rec;
    g_walk_steps = &rec;
    for (size_t i = 0; i < n; ++i) {
        rec.clear();
        g_rng = RngCtx{};
        uint32_t x = (uint32_t)pixels[2 * i], y = (uint32_t)pixels[2 * i + 1];
        g_rng.key = path_key(seed, (uint64_t)x + (uint64_t)y * width, sample_index);
        double u = ((double)x + rand_scalar(0.0, 1.0, MRT_DIM_JITTER_U)) / ((double)width - 1.0);
        double v = ((double)y + rand_scalar(0.0, 1.0, MRT_DIM_JITTER_V)) / ((double)height - 1.0);
        (void)ray_color(s->world.camera->get_ray(u, v), s->world, 50);
        for (uint32_t k = 0; k < max_segments; ++k) steps_out[i * max_segments + k] = k < rec.size() ? rec[k] : 0xFFFFFFFFu;
    }
    g_walk_steps = nullptr;
    return MRT_OK;
}

// Probe of LambertianPDF::generate (pdf.rs:234-282) on n given surface points: point i has normal
// normals[i], the incoming ray's time in_times[i], and draws from the stream of pixel index i, sample 0,
// segment `seg`. Outputs: the sampled direction, the mixture "pdf" value, the branch drawn (0 cosine,
// 1 light list, 2 sky) and whether generate returned Some. For an independent restatement to be checked
// against (tests/test_pdf_restatement.py).
int oracle_lambertian_generate(void* scene, const double* points, const double* normals, const double* in_times, size_t n,
                               uint64_t seed, uint32_t seg, double* dir_out, double* value_out, int32_t* choice_out,
                               int32_t* some_out) {
    auto* s = static_cast<Scene*>(scene);
    if (!s || !points || !normals || !in_times || !dir_out || !value_out || !choice_out || !some_out) return MRT_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; ++i) {
        g_rng = RngCtx{};
        g_rng.key = path_key(seed, (uint64_t)i, 0);
        g_rng.seg = seg;
        const V3 p{points[3 * i], points[3 * i + 1], points[3 * i + 2]};
        const V3 nrm{normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
        const Ray incoming{p, V3{0.0, 0.0, 1.0}, in_times[i]};  // generate reads only the time of the incoming ray
        choice_out[i] = (int32_t)rand_u32(0, 3, MRT_DIM_PDF_CHOICE);
        auto v = lambertian_pdf_generate(nrm, incoming, p, s->world);
        some_out[i] = v ? 1 : 0;
        dir_out[3 * i] = v ? v->direction.x : 0.0, dir_out[3 * i + 1] = v ? v->direction.y : 0.0, dir_out[3 * i + 2] = v ? v->direction.z : 0.0;
        value_out[i] = v ? v->value : 0.0;
    }
    return MRT_OK;
}

uint64_t oracle_rng_u64(uint64_t seed, uint64_t pixel, uint64_t sample, uint32_t seg, uint32_t dim) {
    uint64_t key = path_key(seed, pixel, sample);
    return mix64(key + 0x9E3779B97F4A7C15ULL * ((((uint64_t)seg << 16) | dim) + 1));
}

// ParallelImage::to_image colour mapping (parallel_image.rs:35-65, rgb_color.rs:75-80)
int oracle_to_image(const float* accum_rgb, uint32_t width, uint32_t height, uint32_t num_samples,
                    uint8_t* rgb8_out) {
    auto conv = [](float c) {
        float s = std::sqrt(c);
        if (s < 0.0f) s = 0.0f; else if (s > 1.0f) s = 1.0f;  // prelude.rs:10-18 clamp (NaN passes through)
        float r = std::round(s * 255.0f);
        if (std::isnan(r)) return (uint8_t)0;  // Rust `as u8` of NaN is 0
        return (uint8_t)r;
    };
    for (uint32_t x = 0; x < width; ++x)
        for (uint32_t y = 0; y < height; ++y) {
            const float* p = accum_rgb + ((size_t)x + (size_t)y * width) * 3;
            uint8_t* o = rgb8_out + ((size_t)x + (size_t)(height - y - 1) * width) * 3;
            for (int c = 0; c < 3; ++c) o[c] = conv(p[c] / (float)num_samples);
        }
    return MRT_OK;
}

// Aabb::approx_eq (bvh.rs:56-70): the six absolute differences sum to less than 0.001.
int oracle_aabb_approx_eq(const double* a_min, const double* a_max, const double* b_min, const double* b_max) {
    double total = 0.0;
    for (int i = 0; i < 3; ++i) total += std::fabs(a_min[i] - b_min[i]) + std::fabs(a_max[i] - b_max[i]);
    return total < 0.001 ? 1 : 0;
}

}  // extern "C"
