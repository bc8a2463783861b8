// mrt_post.hip -- the reference's post-processing step on gfx950: bloom (cpu_rt/src/ray_tracer/
// bloom.rs:75-94 with the image operations of cpu_rt/src/parallel_image.rs) and the 8-bit colour
// mapping (parallel_image.rs:35-65, prelude/rgb_color.rs:75-80), on device-resident images.
//
// Images are float[height][width][3] (index (x + y*width)*3 + c), as the accumulation buffer.
// Every kernel is a streaming f32 pass (HBM/L2 bound, one thread per pixel); the arithmetic keeps
// the reference's operation order and is compiled without FMA contraction, so everything up to
// the final exp / pow is bit-identical to an f32 CPU evaluation.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "mrt_internal.h"

#pragma clang fp contract(off)

namespace mrt_post {

constexpr int BLOCK = 256;
constexpr int LEVELS = 6;  // GaussianBlur { amount: 6 }, bloom.rs:81

struct Rgb {
    float r, g, b;
};
__device__ __forceinline__ Rgb operator+(Rgb a, Rgb b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
__device__ __forceinline__ Rgb operator*(Rgb a, float s) { return {a.r * s, a.g * s, a.b * s}; }
__device__ __forceinline__ Rgb operator*(float s, Rgb a) { return a * s; }
__device__ __forceinline__ Rgb ld(const float* img, int w, int x, int y) {
    const float* p = img + ((size_t)x + (size_t)y * w) * 3;
    return {p[0], p[1], p[2]};
}
__device__ __forceinline__ void st(float* img, int w, int x, int y, Rgb c) {
    float* p = img + ((size_t)x + (size_t)y * w) * 3;
    p[0] = c.r, p[1] = c.g, p[2] = c.b;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// original = accum / num_samples (parallel_image.rs:268-278); bright = SelectMinMag { 1.0 } (bloom.rs:60-74)
__global__ void __launch_bounds__(BLOCK) k_select(const float* accum, float num_samples, float* original, float* bright,
                                                  uint32_t n_pixels) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_pixels) return;
    Rgb c{accum[3 * (size_t)i] / num_samples, accum[3 * (size_t)i + 1] / num_samples, accum[3 * (size_t)i + 2] / num_samples};
    original[3 * (size_t)i] = c.r, original[3 * (size_t)i + 1] = c.g, original[3 * (size_t)i + 2] = c.b;
    float mag = c.r * c.r + c.g * c.g + c.b * c.b;
    if (mag <= 1.0f) c = Rgb{0.f, 0.f, 0.f};
    bright[3 * (size_t)i] = c.r, bright[3 * (size_t)i + 1] = c.g, bright[3 * (size_t)i + 2] = c.b;
}

// One pass of GaussianBlur::blur (bloom.rs:10-39): nine clamped taps along x (DY = 0) or y (DY = 1),
// accumulated centre, +1, -1, +2, -2, ... as the reference does.
template <int DY>
__global__ void __launch_bounds__(BLOCK) k_blur(const float* in, float* out, int w, int h) {
    const float wt[5] = {(float)0.227027, (float)0.1945946, (float)0.1216216, (float)0.054054, (float)0.016216};
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (uint32_t)w * (uint32_t)h) return;
    int x = (int)(i % (uint32_t)w), y = (int)(i / (uint32_t)w);
    Rgb result = ld(in, w, x, y) * wt[0];
#pragma unroll
    for (int k = 1; k < 5; ++k) {
        if (DY == 0) {
            result = result + ld(in, w, clampi(x + k, 0, w - 1), y) * wt[k];
            result = result + ld(in, w, clampi(x - k, 0, w - 1), y) * wt[k];
        } else {
            result = result + ld(in, w, x, clampi(y + k, 0, h - 1)) * wt[k];
            result = result + ld(in, w, x, clampi(y - k, 0, h - 1)) * wt[k];
        }
    }
    st(out, w, x, y, result);
}

// ParallelImage::down_sample, parallel_image.rs:193-213
__global__ void __launch_bounds__(BLOCK) k_down_sample(const float* in, int w, int h, float* out, int nw, int nh) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (uint32_t)nw * (uint32_t)nh) return;
    int x = (int)(i % (uint32_t)nw), y = (int)(i / (uint32_t)nw);
    int x1 = min(x * 2 + 1, w - 1), y1 = min(y * 2 + 1, h - 1);
    Rgb s = ld(in, w, x * 2, y * 2) + ld(in, w, x1, y * 2);
    s = s + ld(in, w, x1, y1);
    s = s + ld(in, w, x * 2, y1);
    st(out, nw, x, y, Rgb{s.r / 4.0f, s.g / 4.0f, s.b / 4.0f});
}

// impl Add<&ParallelImage> (parallel_image.rs:246-260) with get_nearest (:215-240): dst += bilinear(src)
__global__ void __launch_bounds__(BLOCK) k_up_add(float* dst, int w, int h, const float* src, int sw, int sh) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (uint32_t)w * (uint32_t)h) return;
    int x = (int)(i % (uint32_t)w), y = (int)(i / (uint32_t)w);
    float y_f = ((float)y + 0.5f) / ((float)h - 1.0f);
    float x_f = ((float)x + 0.5f) / ((float)w - 1.0f);
    float x_pixel = fmaxf(fminf(x_f * (float)(sw - 1), (float)sw - 1.0f), 0.0f);
    float y_pixel = fmaxf(fminf(y_f * (float)(sh - 1), (float)sh - 1.0f), 0.0f);
    int x0 = (int)floorf(x_pixel), y0 = (int)floorf(y_pixel);
    int x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
    Rgb x0y0 = ld(src, sw, x0, y0), x1y0 = ld(src, sw, x1, y0), x1y1 = ld(src, sw, x1, y1), x0y1 = ld(src, sw, x0, y1);
    float x_fract = x_pixel - truncf(x_pixel), y_fract = y_pixel - truncf(y_pixel);
    Rgb c0 = (1.0f - y_fract) * x0y0 + y_fract * x0y1;
    Rgb c1 = (1.0f - y_fract) * x1y0 + y_fract * x1y1;
    Rgb p = (1.0f - x_fract) * c0 + x_fract * c1;
    st(dst, w, x, y, ld(dst, w, x, y) + p);
}

__device__ __forceinline__ float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// bloom.rs:84-93: 1 - exp(-hdr * (bloom + hdr)), then pow(1 / 2.2)
__global__ void __launch_bounds__(BLOCK) k_tone_map(const float* original, const float* bloom_tex, float* out, uint32_t n_values) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_values) return;
    const float inv_gamma = 1.0f / 2.2f;
    float hdr = clamp01(original[i]);
    float bloom_color = bloom_tex[i] + hdr;
    float e = (-1.0f * hdr) * bloom_color;
    out[i] = powf(1.0f - expf(e), inv_gamma);
}

// RgbColor::as_rgb_u8 of image / num_samples with the row flip of to_image (parallel_image.rs:55-63)
__global__ void __launch_bounds__(BLOCK) k_to_rgb8(const float* img, float num_samples, int w, int h, uint8_t* out) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (uint32_t)w * (uint32_t)h) return;
    int x = (int)(i % (uint32_t)w), y = (int)(i / (uint32_t)w);
    const float* p = img + 3 * (size_t)i;
    uint8_t* o = out + ((size_t)x + (size_t)(h - 1 - y) * w) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = sqrtf(p[c] / num_samples);
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);  // prelude clamp: NaN passes through
        float r = roundf(v * 255.0f);
        o[c] = (r != r) ? (uint8_t)0 : (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));  // Rust `as u8`: saturating, NaN -> 0
    }
}

struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    float* get(size_t n_floats) {
        void* p = nullptr;
        if (hipMalloc(&p, sizeof(float) * (n_floats ? n_floats : 4)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return static_cast<float*>(p);
    }
};

// Work images of mrt_bloom_device, kept per device between calls (an interactive caller blooms every
// frame at one size; allocating and freeing ten buffers per call cost more than the kernels). Replaced
// when the image size changes; calls on one device take turns.
struct BloomWork {
    std::mutex mu;
    uint32_t width = 0, height = 0;
    std::unique_ptr<Scratch> mem;
    float *original = nullptr, *tmp_a = nullptr, *tmp_b = nullptr;
    float* level[LEVELS + 1] = {};
};
constexpr int MAX_DEVICES = 64;
// Never destroyed: the HIP runtime registers its own teardown at the first HIP call, i.e. after this
// library was loaded, so at process exit it is gone before a static destructor here would run, and
// hipFree on a torn-down runtime is undefined. The process's end returns the memory.
BloomWork* const g_bloom_work = new BloomWork[MAX_DEVICES];

inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }

}  // namespace mrt_post

using namespace mrt_post;

extern "C" {

int mrt_bloom_device(int device, const float* accum_rgb_device, uint32_t width, uint32_t height, uint32_t num_samples,
                     float* out_rgb_device, void* stream_) {
    auto fail = [&](int code, const std::string& m) {
        mrt::set_last_error(m);
        return code;
    };
    if (!accum_rgb_device || !out_rgb_device) return fail(MRT_ERR_INVALID_ARGUMENT, "NULL image");
    if (num_samples == 0) return fail(MRT_ERR_INVALID_ARGUMENT, "num_samples is zero");
    if ((uint64_t)width * height > 0xFFFFFFFFull / 3) return fail(MRT_ERR_INVALID_ARGUMENT, "image too large");
    // the reference panics when the last pyramid level is narrower than two pixels (parallel_image.rs:216-217)
    if ((width >> LEVELS) < 2 || (height >> LEVELS) < 2)
        return fail(MRT_ERR_UNSUPPORTED, "bloom needs an image of at least 128x128 (the reference panics below that)");
    if (device < 0 || device >= MAX_DEVICES) return fail(MRT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    if (hipSetDevice(device) != hipSuccess) return fail(MRT_ERR_DEVICE, "hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    BloomWork& work = g_bloom_work[device];
    std::lock_guard<std::mutex> turn(work.mu);
    int w[LEVELS + 1], h[LEVELS + 1];
    w[0] = (int)width, h[0] = (int)height;
    for (int i = 1; i <= LEVELS; ++i) w[i] = w[i - 1] / 2, h[i] = h[i - 1] / 2;
    const size_t n0 = (size_t)width * height;
    if (!work.mem || work.width != width || work.height != height) {
        work.mem.reset(new Scratch());
        work.width = work.height = 0;
        work.original = work.mem->get(3 * n0);
        work.tmp_a = work.mem->get(3 * n0);  // bright image / down-sampled input
        work.tmp_b = work.mem->get(3 * n0);  // x-blurred
        bool ok = work.original && work.tmp_a && work.tmp_b;
        for (int i = 0; i <= LEVELS; ++i) ok = (work.level[i] = work.mem->get(3 * (size_t)w[i] * h[i])) != nullptr && ok;
        if (!ok) {
            work.mem.reset();
            return fail(MRT_ERR_OUT_OF_MEMORY, "hipMalloc failed");
        }
        work.width = width, work.height = height;
    }
    float *original = work.original, *tmp_a = work.tmp_a, *tmp_b = work.tmp_b;
    float** level = work.level;

    hipLaunchKernelGGL(k_select, grid_for(n0), dim3(BLOCK), 0, stream, accum_rgb_device, (float)num_samples, original, tmp_a,
                       (uint32_t)n0);
    for (int i = 0; i <= LEVELS; ++i) {
        const size_t n = (size_t)w[i] * h[i];
        if (i > 0)
            hipLaunchKernelGGL(k_down_sample, grid_for(n), dim3(BLOCK), 0, stream, level[i - 1], w[i - 1], h[i - 1], tmp_a, w[i],
                               h[i]);
        hipLaunchKernelGGL(k_blur<0>, grid_for(n), dim3(BLOCK), 0, stream, tmp_a, tmp_b, w[i], h[i]);
        hipLaunchKernelGGL(k_blur<1>, grid_for(n), dim3(BLOCK), 0, stream, tmp_b, level[i], w[i], h[i]);
    }
    for (int i = LEVELS - 1; i >= 0; --i)
        hipLaunchKernelGGL(k_up_add, grid_for((size_t)w[i] * h[i]), dim3(BLOCK), 0, stream, level[i], w[i], h[i], level[i + 1],
                           w[i + 1], h[i + 1]);
    hipLaunchKernelGGL(k_tone_map, grid_for(3 * n0), dim3(BLOCK), 0, stream, original, level[0], out_rgb_device,
                       (uint32_t)(3 * n0));
    if (hipGetLastError() != hipSuccess) return fail(MRT_ERR_DEVICE, "kernel launch failed");
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(MRT_ERR_DEVICE, "bloom kernels failed");
    return MRT_OK;
}

int mrt_to_image_device(int device, const float* rgb_device, uint32_t width, uint32_t height, uint32_t num_samples,
                        uint8_t* rgb8_out_device, void* stream_) {
    auto fail = [&](int code, const std::string& m) {
        mrt::set_last_error(m);
        return code;
    };
    if (!rgb_device || !rgb8_out_device) return fail(MRT_ERR_INVALID_ARGUMENT, "NULL image");
    if (num_samples == 0) return fail(MRT_ERR_INVALID_ARGUMENT, "num_samples is zero");
    if ((uint64_t)width * height > 0xFFFFFFFFull / 3) return fail(MRT_ERR_INVALID_ARGUMENT, "image too large");
    if (hipSetDevice(device) != hipSuccess) return fail(MRT_ERR_DEVICE, "hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (width == 0 || height == 0) return MRT_OK;
    hipLaunchKernelGGL(k_to_rgb8, grid_for((size_t)width * height), dim3(BLOCK), 0, stream, rgb_device, (float)num_samples,
                       (int)width, (int)height, rgb8_out_device);
    if (hipGetLastError() != hipSuccess) return fail(MRT_ERR_DEVICE, "kernel launch failed");
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(MRT_ERR_DEVICE, "colour mapping kernel failed");
    return MRT_OK;
}

int mrt_bloom(int device, const float* accum_rgb, uint32_t width, uint32_t height, uint32_t num_samples, float* out_rgb) {
    auto fail = [&](int code, const std::string& m) {
        mrt::set_last_error(m);
        return code;
    };
    if (!accum_rgb || !out_rgb) return fail(MRT_ERR_INVALID_ARGUMENT, "NULL image");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(MRT_ERR_DEVICE, "no HIP device is visible; the HIP path is required (there is no CPU fallback)");
    if (device < 0 || device >= n_dev) return fail(MRT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    if (hipSetDevice(device) != hipSuccess) return fail(MRT_ERR_DEVICE, "hipSetDevice failed");
    const size_t bytes = sizeof(float) * 3 * (size_t)width * height;
    Scratch mem;
    float* d_in = mem.get(3 * (size_t)width * height);
    float* d_out = mem.get(3 * (size_t)width * height);
    if (!d_in || !d_out) return fail(MRT_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    if (hipMemcpy(d_in, accum_rgb, bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(MRT_ERR_DEVICE, "upload failed");
    int rc = mrt_bloom_device(device, d_in, width, height, num_samples, d_out, nullptr);
    if (rc != MRT_OK) return rc;
    if (hipMemcpy(out_rgb, d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(MRT_ERR_DEVICE, "download failed");
    return MRT_OK;
}

}  // extern "C"
