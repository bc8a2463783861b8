"""The render entry point that allocates nothing (mrt_render_accumulate_device_noalloc) against
mrt_render_accumulate_device: same samples, same counters, images equal up to the order of the f32
atomic additions (rtol 1e-5, atol 1e-6: the bar of every other test that compares two schedules of
one render). Two calls in a row on one GPU, the second smaller, so the second call meets the first
call's leftovers in the library's scratch array (pool slots beyond its own, counters, work ranges)."""
import ctypes as C

import numpy as np
import pytest

from ray_tracing_amd import RayTracer, _ffi as F, scenarios

pytestmark = pytest.mark.gpu


def _render(lib, fn, rt, dev_buf, w, h, spp, first, x0=0, strip_w=None):
    import torch
    st = F.Stats()
    dev_buf.zero_()
    torch.cuda.synchronize()
    rc = fn(rt.world.handle, dev_buf.data_ptr(), w, h, x0, w - x0 if strip_w is None else strip_w, spp, rt.seed, first,
            None, C.byref(st))
    assert rc == F.MRT_OK, lib.mrt_last_error()
    return dev_buf.cpu().numpy().copy(), st


@pytest.mark.parametrize("name", ["Cornell Smoke", "Random Scene", "Fast Oct Tree Sphere"])
def test_noalloc_matches_allocating_entry_point_twice(name):
    import torch
    from tests.helpers import scene_by_name
    lib = F.load()
    rt = RayTracer(scene_by_name(name), device=0, seed=5)
    # first call: more paths than the scratch pool has slots (524,288), so every slot is used and regenerated
    calls = [(640, 360, 3, 2, 0, None), (97, 61, 2, 7, 5, 40)]
    for w, h, spp, first, x0, sw in calls:
        buf = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
        got, st = _render(lib, lib.mrt_render_accumulate_device_noalloc, rt, buf, w, h, spp, first, x0, sw)
        ref, rst = _render(lib, lib.mrt_render_accumulate_device, rt, buf, w, h, spp, first, x0, sw)
        assert st.paths == rst.paths == (w - x0 if sw is None else sw) * h * spp
        assert st.rays == rst.rays and st.light_queries == rst.light_queries
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
        if sw is not None:
            assert (got[:, :x0] == 0).all() and (got[:, x0 + sw:] == 0).all()


def test_noalloc_single_bounce_shader_and_argument_checks():
    import torch
    lib = F.load()
    rt = RayTracer(scenarios.cornell_smoke(), device=0, seed=2)
    rt.set_shader("Diffuse")
    buf = torch.zeros((45, 80, 3), dtype=torch.float32, device="cuda:0")
    got, st = _render(lib, lib.mrt_render_accumulate_device_noalloc, rt, buf, 80, 45, 2, 0)
    ref, rst = _render(lib, lib.mrt_render_accumulate_device, rt, buf, 80, 45, 2, 0)
    assert st.rays == rst.rays
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    rt.set_shader("Ray Tracing")
    fn = lib.mrt_render_accumulate_device_noalloc
    h = rt.world.handle
    assert fn(h, buf.data_ptr(), 80, 45, 70, 20, 1, 0, 0, None, None) == F.MRT_ERR_INVALID_ARGUMENT   # strip outside
    assert fn(h, None, 80, 45, 0, 80, 1, 0, 0, None, None) == F.MRT_ERR_INVALID_ARGUMENT
    assert fn(h, buf.data_ptr(), 1, 45, 0, 1, 1, 0, 0, None, None) == F.MRT_ERR_INVALID_ARGUMENT
    assert fn(None, buf.data_ptr(), 80, 45, 0, 80, 1, 0, 0, None, None) == F.MRT_ERR_INVALID_ARGUMENT
    host = np.zeros((45, 80, 3), dtype=np.float32)   # host memory is not a device accumulation buffer
    assert fn(h, host.ctypes.data, 80, 45, 0, 80, 1, 0, 0, None, None) == F.MRT_ERR_INVALID_ARGUMENT
    assert lib.mrt_render_accumulate_device(h, host.ctypes.data, 80, 45, 0, 80, 1, 0, 0, None, None) == F.MRT_ERR_INVALID_ARGUMENT


def test_noalloc_leaves_the_free_device_memory_unchanged():
    """On a scene that has never rendered (so the allocating entry point's pool does not exist yet) the call must not
    change the device's free memory, on the first call or on a second, smaller one. A warm-up render of ANOTHER scene
    handle comes first: the runtime itself reserves memory the first time a kernel with a private segment runs on a
    queue, which is no allocation of this library's."""
    import torch
    lib = F.load()
    fn = lib.mrt_render_accumulate_device_noalloc
    info = scenarios.cornell_smoke()
    warm = RayTracer(info, device=0, seed=1)
    buf = torch.zeros((360, 640, 3), dtype=torch.float32, device="cuda:0")
    small = torch.zeros((61, 97, 3), dtype=torch.float32, device="cuda:0")
    st = F.Stats()
    assert fn(warm.world.handle, buf.data_ptr(), 640, 360, 0, 640, 1, 1, 0, None, C.byref(st)) == F.MRT_OK
    fresh = RayTracer(info, device=0, seed=5)      # created (scene data uploaded), never rendered
    torch.cuda.synchronize()
    for tensor, w, h, spp in ((buf, 640, 360, 3), (small, 97, 61, 2)):
        free_before, _ = torch.cuda.mem_get_info(0)
        assert fn(fresh.world.handle, tensor.data_ptr(), w, h, 0, w, spp, 5, 0, None, C.byref(st)) == F.MRT_OK, lib.mrt_last_error()
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info(0)
        assert free_after == free_before, (free_before, free_after)
        assert st.paths == w * h * spp
    # the allocating entry point, by contrast, makes the scene's pool on its first call
    free_before, _ = torch.cuda.mem_get_info(0)
    assert lib.mrt_render_accumulate_device(fresh.world.handle, buf.data_ptr(), 640, 360, 0, 640, 1, 5, 0, None, C.byref(st)) == F.MRT_OK
    free_after, _ = torch.cuda.mem_get_info(0)
    assert free_after < free_before
