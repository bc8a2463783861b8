"""Ad-hoc GPU sanity run: device path vs CPU checker on Cornell smoke."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import scene_by_name
from ray_tracing_amd import scenarios, RayTracer, ParallelImage, World
from tests.oracle_ffi import OracleWorld

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "Cornell Smoke"
    info = scene_by_name(name)
    ow = OracleWorld(info)
    rt = RayTracer(info, device=0, seed=7)
    gw = rt.world
    bo, bg = ow.bvh(), gw.bvh()
    print("bvh equal:", len(bo) == len(bg) and all((bo[k] == bg[k]).all() for k in bo.dtype.names))
    W = H = 96
    ys, xs = np.mgrid[0:H, 0:W]
    px = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    ro = ow.camera_rays(W, H, px, 7, 3)
    rg = gw.camera_rays(W, H, px, 7, 3)
    print("camera max abs diff:", np.abs(ro - rg).max())
    oo, to, no, po, fo = ow.trace_rays(ro, seed=11)
    og, tg, ng, pg, fg = gw.trace_rays(ro, seed=11)
    print("primary: id mismatches", int((oo != og).sum()), "of", len(oo), "hits", int((oo >= 0).sum()))
    m = (oo == og) & (oo >= 0)
    print(" t rel err", np.abs(to[m] - tg[m]).max() / np.abs(to[m]).max(), "normal err", np.abs(no[m] - ng[m]).max(),
          "pos err", np.abs(po[m] - pg[m]).max(), "ff mism", int((fo[m] != fg[m]).sum()))
    spp = 8
    t0 = time.time(); ao, so = ow.render(W, H, spp, seed=7, threads=16); t1 = time.time()
    img = ParallelImage.new_black(W, H)
    rt.world.set_option("trace_zero_throughput", 1)
    st = rt.trace_part(img.split(1)[0], spp, first_sample=0); t2 = time.time()
    ag = img.buffer
    d = np.abs(ao - ag)
    print("render: oracle rays", so.rays, "lq", so.light_queries, "gpu rays", st.rays, "lq", st.light_queries)
    print(" max abs diff", d.max(), "pixels differing >1e-3:", int((d.max(axis=2) > 1e-3 * spp).sum()), "of", W * H,
          "mean o", ao.mean() / spp, "mean g", ag.mean() / spp)
    print(" oracle s", t1 - t0, "gpu s", t2 - t1, "kernel ms", st.kernel_ms, "launches", st.launches)
    # throughput probe
    rt.world.set_option("trace_zero_throughput", 0)
    for (w, h, s) in [(960, 540, 16), (1920, 1080, 16)]:
        img = ParallelImage.new_black(w, h)
        t0 = time.time(); st = rt.trace_part(img.split(1)[0], s, first_sample=0); t1 = time.time()
        print(f" {w}x{h}x{s}: total_ms {st.total_ms:.1f} kernel_ms {st.kernel_ms:.1f} launches {st.launches} "
              f"Mpaths/s {st.paths / st.total_ms / 1e3:.1f} Mrays/s {st.rays / st.total_ms / 1e3:.1f} wall {t1 - t0:.2f}")

main()
