"""Post-processing step (bloom + 8-bit colour mapping): the CPU restatement against an independent
numpy restatement of cpu_rt/src/ray_tracer/bloom.rs + cpu_rt/src/parallel_image.rs (CPU tests), and
the HIP kernels against the CPU restatement through the C ABI (gpu tests).

The reference holds no test vectors for this code: parity unpinned by the reference; the two
restatements were written separately (loops in C++, whole-array numpy here) from the cited lines.
"""
import numpy as np
import pytest

from tests.oracle_ffi import oracle_bloom, oracle_to_image

F32 = np.float32
WEIGHTS = [F32(0.227027), F32(0.1945946), F32(0.1216216), F32(0.054054), F32(0.016216)]


def np_blur(img):
    """GaussianBlur::blur, bloom.rs:10-39 (img: (H, W, 3) f32, index [y, x])."""
    h, w, _ = img.shape

    def pass_(src, axis):
        n = src.shape[axis]
        idx = np.arange(n)
        res = src * WEIGHTS[0]
        for i in range(1, 5):
            res = res + np.take(src, np.clip(idx + i, 0, n - 1), axis=axis) * WEIGHTS[i]
            res = res + np.take(src, np.clip(idx - i, 0, n - 1), axis=axis) * WEIGHTS[i]
        return res.astype(F32)

    return pass_(pass_(img, 1), 0)


def np_down_sample(img):  # parallel_image.rs:193-213
    h, w, _ = img.shape
    nh, nw = h // 2, w // 2
    x0, y0 = np.arange(nw) * 2, np.arange(nh) * 2
    x1, y1 = np.minimum(x0 + 1, w - 1), np.minimum(y0 + 1, h - 1)
    s = img[np.ix_(y0, x0)] + img[np.ix_(y0, x1)]
    s = s + img[np.ix_(y1, x1)]
    s = s + img[np.ix_(y1, x0)]
    return (s / F32(4.0)).astype(F32)


def np_up_add(dst, src):  # parallel_image.rs:215-260
    h, w, _ = dst.shape
    sh, sw, _ = src.shape
    y_f = (np.arange(h, dtype=F32) + F32(0.5)) / (F32(h) - F32(1.0))
    x_f = (np.arange(w, dtype=F32) + F32(0.5)) / (F32(w) - F32(1.0))
    xp = np.maximum(np.minimum(x_f * F32(sw - 1), F32(sw) - F32(1.0)), F32(0.0)).astype(F32)
    yp = np.maximum(np.minimum(y_f * F32(sh - 1), F32(sh) - F32(1.0)), F32(0.0)).astype(F32)
    x0, y0 = np.floor(xp).astype(np.int64), np.floor(yp).astype(np.int64)
    x1, y1 = np.minimum(x0 + 1, sw - 1), np.minimum(y0 + 1, sh - 1)
    xf = (xp - np.trunc(xp)).astype(F32)[None, :, None]
    yf = (yp - np.trunc(yp)).astype(F32)[:, None, None]
    c0 = (F32(1.0) - yf) * src[np.ix_(y0, x0)] + yf * src[np.ix_(y1, x0)]
    c1 = (F32(1.0) - yf) * src[np.ix_(y0, x1)] + yf * src[np.ix_(y1, x1)]
    return (dst + ((F32(1.0) - xf) * c0 + xf * c1)).astype(F32)


def np_bloom(accum, num_samples):  # bloom.rs:75-94 after ray_tracer.rs:473
    original = (accum / F32(num_samples)).astype(F32)
    mag = original[..., 0] ** 2 + original[..., 1] ** 2 + original[..., 2] ** 2
    bright = np.where((mag <= F32(1.0))[..., None], F32(0.0), original).astype(F32)
    levels = [np_blur(bright)]
    for _ in range(6):
        levels.append(np_blur(np_down_sample(levels[-1])))
    for i in range(5, -1, -1):
        levels[i] = np_up_add(levels[i], levels[i + 1])
    hdr = np.clip(original, F32(0.0), F32(1.0))
    bloom_color = levels[0] + hdr
    e = (F32(-1.0) * hdr) * bloom_color
    return np.power(F32(1.0) - np.exp(e, dtype=F32), F32(1.0) / F32(2.2), dtype=F32)


def assert_tone_mapped_close(got, want):
    """Compares two bloom outputs. The last two steps are x = 1 - exp(e) and x ** (1 / 2.2). exp(e)
    is a value near 1 for dim pixels, so one ulp of difference between two exp implementations
    (6e-8) is an ABSOLUTE error of x, which the power then stretches for small x. The comparison is
    therefore made on x (= out ** 2.2) with an absolute tolerance of a few ulps of 1.0 plus the
    north-star relative tolerance, and on the 8-bit image (at most one level)."""
    gx = np.asarray(got, dtype=np.float64) ** 2.2
    wx = np.asarray(want, dtype=np.float64) ** 2.2
    np.testing.assert_allclose(gx, wx, rtol=3e-5, atol=4e-7)
    # 8-bit image: at most one level, except at the black end where sqrt stretches the same noise
    g8, w8 = oracle_to_image(np.asarray(got, np.float32), 1), oracle_to_image(np.asarray(want, np.float32), 1)
    d8 = np.abs(g8.astype(int) - w8.astype(int))
    bright_enough = (wx > 1e-4)[::-1]          # to_image flips the rows
    assert d8[bright_enough].max(initial=0) <= 1
    assert d8.max() <= 8


def synthetic_accum(h, w, spp, seed):
    """Mostly dim radiance with a few bright emitters and fireflies (values above the bright-pass
    threshold), scaled like an accumulation of `spp` samples."""
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 3), dtype=np.float32) * F32(0.6)
    img[h // 3: h // 3 + 9, w // 4: w // 4 + 40] = F32(7.0)
    ys, xs = rng.integers(0, h, 50), rng.integers(0, w, 50)
    img[ys, xs] = rng.random((50, 3), dtype=np.float32) * F32(30.0)
    return (img * F32(spp)).astype(np.float32)


@pytest.mark.parametrize("shape", [(128, 128), (130, 257), (270, 480)])
def test_cpu_restatements_agree(shape):
    h, w = shape
    accum = synthetic_accum(h, w, 4, seed=h + w)
    a = oracle_bloom(accum, 4)
    b = np_bloom(accum, 4)
    assert a is not None
    # same f32 operations in the same order; exp / pow come from different libms
    assert_tone_mapped_close(a, b)


def test_bloom_rejects_images_the_reference_panics_on():
    assert oracle_bloom(np.zeros((127, 300, 3), np.float32), 1) is None
    assert oracle_bloom(np.zeros((300, 64, 3), np.float32), 1) is None


def test_bloom_of_dim_image_is_pure_tone_map():
    """No pixel passes the bright-pass, so the bloom texture is zero: out = (1 - exp(-c^2))^(1/2.2)."""
    rng = np.random.default_rng(5)
    img = (rng.random((128, 160, 3), dtype=np.float32) * F32(0.5)).astype(np.float32)
    out = oracle_bloom(img, 1)
    want = np.power(F32(1.0) - np.exp(-(img * img), dtype=F32), F32(1.0) / F32(2.2), dtype=F32)
    assert_tone_mapped_close(out, want)


def test_bloom_spreads_a_bright_pixel():
    img = np.zeros((256, 256, 3), np.float32)
    img[128, 128] = 50.0
    out = oracle_bloom(img, 1)
    assert out[128, 128].min() > 0.99        # saturated centre
    assert np.all(out[0, 0] == 0.0)          # hdr = 0 there: exp(-0 * bloom) = 1 whatever the bloom is
    # a dim pixel near the emitter is lifted by the bloom term, a far one is not
    img2 = img.copy()
    img2[128, 140] = 0.2
    img2[10, 10] = 0.2
    out2 = oracle_bloom(img2, 1)
    assert out2[128, 140, 0] > out2[10, 10, 0] > 0.0


# ------------------------------------------------------------------ GPU (through the C ABI)
def _gpu_bloom(accum, spp):
    import ctypes as C
    from ray_tracing_amd import _ffi as F
    lib = F.load()
    out = np.empty_like(accum)
    h, w, _ = accum.shape
    F.check(lib.mrt_bloom(0, accum.ctypes.data, w, h, spp, out.ctypes.data))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 128), (130, 257), (270, 480), (1080, 1920)])
def test_gpu_bloom_matches_cpu_restatement(shape):
    h, w = shape
    accum = synthetic_accum(h, w, 8, seed=h * 3 + w)
    want = oracle_bloom(accum, 8)
    got = _gpu_bloom(accum, 8)
    # identical f32 arithmetic up to exp / pow (device libm vs host libm)
    assert_tone_mapped_close(got, want)


@pytest.mark.gpu
def test_gpu_bloom_error_behaviour():
    from ray_tracing_amd import _ffi as F
    lib = F.load()
    small = np.zeros((100, 300, 3), np.float32)
    out = np.empty_like(small)
    assert lib.mrt_bloom(0, small.ctypes.data, 300, 100, 1, out.ctypes.data) == F.MRT_ERR_UNSUPPORTED
    ok = np.zeros((128, 128, 3), np.float32)
    out = np.empty_like(ok)
    assert lib.mrt_bloom(0, ok.ctypes.data, 128, 128, 0, out.ctypes.data) == F.MRT_ERR_INVALID_ARGUMENT


@pytest.mark.gpu
def test_gpu_device_pipeline_render_bloom_to_image():
    """tracing_loop on device buffers: render -> bloom -> 8-bit image, against the CPU restatements
    applied to the same accumulation buffer."""
    import torch
    from ray_tracing_amd import RayTracer, scenarios
    W, H, spp = 256, 192, 8
    rt = RayTracer(scenarios.cornell_smoke(), device=0, seed=4)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rt.render_device(accum.data_ptr(), W, H, spp)
    bloomed = torch.empty_like(accum)
    rt.bloom_device(accum.data_ptr(), W, H, spp, bloomed.data_ptr())
    rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    rt.to_image_device(bloomed.data_ptr(), W, H, 1, rgb8.data_ptr())
    torch.cuda.synchronize()
    want_bloom = oracle_bloom(accum.cpu().numpy(), spp)
    assert_tone_mapped_close(bloomed.cpu().numpy(), want_bloom)
    want8 = oracle_to_image(bloomed.cpu().numpy(), 1)
    got8 = rgb8.cpu().numpy()
    assert np.abs(got8.astype(int) - want8.astype(int)).max() <= 0   # same f32 sqrt / round on the same input
    # host-buffer mirror: tracing_loop returns the bloomed image
    from ray_tracing_amd import ParallelImage
    rt2 = RayTracer(scenarios.cornell_smoke(), device=0, seed=4)
    img = ParallelImage.new_black(W, H)
    out = rt2.tracing_loop(img, spp)
    # a second render: the f32 accumulation order differs between runs, so only close
    np.testing.assert_allclose(out.buffer.astype(np.float64) ** 2.2, want_bloom.astype(np.float64) ** 2.2, rtol=1e-3, atol=1e-5)
