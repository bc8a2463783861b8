// post_ref.cpp -- CPU restatement of the reference's post-processing step (TEST INFRASTRUCTURE ONLY:
// loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline; never by the product).
//
// Follows cpu_rt/src/ray_tracer/bloom.rs and cpu_rt/src/parallel_image.rs operation by operation in
// f32, in the reference's evaluation order. The reference holds no test vectors for this code:
// parity unpinned (pinned only by being a line-by-line restatement and by an independent numpy
// restatement in tests/test_post.py).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Rgb {
    float r, g, b;
};
inline Rgb operator+(Rgb a, Rgb b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Rgb operator-(Rgb a, Rgb b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
inline Rgb operator*(Rgb a, float s) { return {a.r * s, a.g * s, a.b * s}; }  // rgb_color.rs:128-136
inline Rgb operator*(float s, Rgb a) { return a * s; }                        // rgb_color.rs:148-153
inline Rgb operator*(Rgb a, Rgb b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Rgb operator/(Rgb a, float s) { return {a.r / s, a.g / s, a.b / s}; }
inline float clamp01(float v) {  // f32::clamp(0.0, 1.0): NaN stays NaN
    if (v < 0.0f) return 0.0f;
    if (v > 1.0f) return 1.0f;
    return v;
}

// ParallelImage (parallel_image.rs:19-24): row-major, index x + y * width
struct Image {
    std::vector<Rgb> buffer;
    int width = 0, height = 0;
    Image() = default;
    Image(int w, int h) : buffer((size_t)w * h, Rgb{0.f, 0.f, 0.f}), width(w), height(h) {}
    Rgb get_xy(int x, int y) const { return buffer[(size_t)x + (size_t)y * width]; }
    void set_xy(int x, int y, Rgb c) { buffer[(size_t)x + (size_t)y * width] = c; }
    Rgb get_clamped(int x, int y) const {  // parallel_image.rs:91-96
        int cx = std::min(std::max(x, 0), width - 1), cy = std::min(std::max(y, 0), height - 1);
        return get_xy(cx, cy);
    }
    Image down_sample() const {  // parallel_image.rs:193-213
        Image img(width / 2, height / 2);
        for (int y = 0; y < img.height; ++y)
            for (int x = 0; x < img.width; ++x) {
                Rgb x0y0 = get_xy(x * 2, y * 2);
                Rgb x1y0 = get_xy(std::min(x * 2 + 1, width - 1), y * 2);
                Rgb x1y1 = get_xy(std::min(x * 2 + 1, width - 1), std::min(y * 2 + 1, height - 1));
                Rgb x0y1 = get_xy(x * 2, std::min(y * 2 + 1, height - 1));
                img.set_xy(x, y, (x0y0 + x1y0 + x1y1 + x0y1) / 4.0f);
            }
        return img;
    }
    Rgb get_nearest(float x, float y) const {  // parallel_image.rs:215-240 (bilinear despite the name)
        float x_pixel = std::fmax(std::fmin(x * (float)(width - 1), (float)width - 1.0f), 0.0f);
        float y_pixel = std::fmax(std::fmin(y * (float)(height - 1), (float)height - 1.0f), 0.0f);
        int x0 = (int)std::floor(x_pixel), y0 = (int)std::floor(y_pixel);
        int x1 = std::min(x0 + 1, width - 1), y1 = std::min(y0 + 1, height - 1);
        Rgb x0y0 = get_xy(x0, y0), x1y0 = get_xy(x1, y0), x1y1 = get_xy(x1, y1), x0y1 = get_xy(x0, y1);
        float x_fract = x_pixel - std::trunc(x_pixel), y_fract = y_pixel - std::trunc(y_pixel);  // f32::fract
        Rgb c0 = (1.0f - y_fract) * x0y0 + y_fract * x0y1;
        Rgb c1 = (1.0f - y_fract) * x1y0 + y_fract * x1y1;
        return (1.0f - x_fract) * c0 + x_fract * c1;
    }
    // impl Add<&ParallelImage> for ParallelImage, parallel_image.rs:246-260
    void add_scaled(const Image& rhs) {
        for (int y = 0; y < height; ++y) {
            float y_f = ((float)y + 0.5f) / ((float)height - 1.0f);
            for (int x = 0; x < width; ++x) {
                float x_f = ((float)x + 0.5f) / ((float)width - 1.0f);
                Rgb p = rhs.get_nearest(x_f, y_f);
                Rgb& d = buffer[(size_t)x + (size_t)y * width];
                d = d + p;
            }
        }
    }
};

// GaussianBlur::blur, bloom.rs:10-39. The weights are untyped float literals: with both Mul<f32> and
// Mul<f64> implemented for RgbColor they default to f64 and are narrowed to f32 inside Mul<f64>
// (rgb_color.rs:138-146).
Image blur(const Image& texture) {
    const float w[5] = {(float)0.227027, (float)0.1945946, (float)0.1216216, (float)0.054054, (float)0.016216};
    Image x_mod(texture.width, texture.height);  // the NaN filter of :14 is overwritten for every pixel
    for (int x = 0; x < texture.width; ++x)
        for (int y = 0; y < texture.height; ++y) {
            Rgb result = texture.get_xy(x, y) * w[0];
            for (int i = 1; i < 5; ++i) {
                result = result + texture.get_clamped(x + i, y) * w[i];
                result = result + texture.get_clamped(x - i, y) * w[i];
            }
            x_mod.set_xy(x, y, result);
        }
    Image y_mod(texture.width, texture.height);
    for (int x = 0; x < texture.width; ++x)
        for (int y = 0; y < texture.height; ++y) {
            Rgb result = x_mod.get_xy(x, y) * w[0];
            for (int i = 1; i < 5; ++i) {
                result = result + x_mod.get_clamped(x, y + i) * w[i];
                result = result + x_mod.get_clamped(x, y - i) * w[i];
            }
            y_mod.set_xy(x, y, result);
        }
    return y_mod;
}

}  // namespace

extern "C" {

// Smallest image side for which the reference's bloom does not panic: the seventh pyramid level
// must be at least 2 pixels wide and high (get_nearest asserts, parallel_image.rs:216-217).
int oracle_bloom_min_side(void) { return 128; }

// tracing_loop's tail (ray_tracer.rs:473-476): image = bloom(accum / num_samples), in place.
// rgb: float[height][width][3]. Returns 0, or -1 when the reference would panic.
int oracle_bloom(float* rgb, uint32_t width, uint32_t height, uint32_t num_samples) {
    if ((width >> 6) < 2 || (height >> 6) < 2) return -1;
    Image original((int)width, (int)height);
    std::memcpy(original.buffer.data(), rgb, sizeof(float) * 3 * (size_t)width * height);
    for (Rgb& c : original.buffer) c = c / (float)num_samples;  // parallel_image.rs:268-278
    // SelectMinMag { min_mag: 1.0 }, bloom.rs:60-74
    Image bright = original;
    for (Rgb& c : bright.buffer)
        if (c.r * c.r + c.g * c.g + c.b * c.b <= 1.0f) c = Rgb{0.f, 0.f, 0.f};
    // GaussianBlur { amount: 6 }, bloom.rs:41-56
    std::vector<Image> down;
    down.push_back(blur(bright));
    for (int i = 0; i < 6; ++i) down.push_back(blur(down.back().down_sample()));
    for (int i = 5; i >= 0; --i) down[(size_t)i].add_scaled(down[(size_t)i + 1]);
    const Image& bloom_texture = down[0];
    // bloom.rs:84-93
    const float gamma = 2.2f;
    const float inv_gamma = 1.0f / gamma;
    for (size_t i = 0; i < original.buffer.size(); ++i) {
        Rgb o = original.buffer[i];
        Rgb hdr{clamp01(o.r), clamp01(o.g), clamp01(o.b)};
        Rgb bloom_color = bloom_texture.buffer[i] + hdr;
        Rgb e = (-1.0f * hdr) * bloom_color;
        Rgb set{1.0f - std::exp(e.r), 1.0f - std::exp(e.g), 1.0f - std::exp(e.b)};
        rgb[3 * i + 0] = std::pow(set.r, inv_gamma);
        rgb[3 * i + 1] = std::pow(set.g, inv_gamma);
        rgb[3 * i + 2] = std::pow(set.b, inv_gamma);
    }
    return 0;
}

}  // extern "C"
