"""One alignment pair end to end through matcher.section_matcher: two S x S uint8 sections (the second resampled through a
known smooth field), irregular meshes of `--mesh-size` px, spacings 280 / 70 px (0.7 x the [400, 100] of
alignment_configs.yaml:16-23).  Prints rounds, timing and the error of the recovered field."""
import argparse
import time

import numpy as np
from scipy.ndimage import gaussian_filter, map_coordinates
from scipy.spatial import Delaunay

from feabas_amd import matcher, renderer
from feabas_amd.mesh import Mesh


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=8192)
    ap.add_argument('--mesh-size', type=float, default=100.0)
    ap.add_argument('--profile', action='store_true', help='cProfile of a third repetition')
    ap.add_argument('--kernels', action='store_true', help='event profile of the kernels of a further repetition')
    ap.add_argument('--entries', action='store_true', help='wall time spent INSIDE every fb_* entry of the library during a further repetition '
                                                           '(what is left of the wall time is python / numpy between the calls)')
    args = ap.parse_args()
    S = args.size
    rng = np.random.default_rng(0)
    t = gaussian_filter(rng.standard_normal((S, S)).astype(np.float32), 1.5)
    t += 0.7 * t.std() / 1.0 * gaussian_filter(rng.standard_normal((S, S)).astype(np.float32), 12) / 0.03
    base = np.clip(128 + 40 * t / t.std(), 0, 255).astype(np.uint8)
    yy, xx = np.meshgrid(np.arange(S, dtype=np.float32), np.arange(S, dtype=np.float32), indexing='ij')

    def field(x, y):
        return (8.0 * np.sin(2 * np.pi * y / (0.8 * S) + 0.4) + 3.0 * (x / S) ** 2, 6.0 * np.cos(2 * np.pi * x / (0.7 * S)) - 2.0 * (x / S) * (y / S))
    ux, uy = field(xx, yy)
    img1 = np.clip(np.rint(map_coordinates(base, [yy + uy, xx + ux], order=1, mode='nearest', output=np.float32)), 0, 255).astype(np.uint8)
    meshes = []
    for k in range(2):
        g = np.arange(0, S, args.mesh_size)
        gx, gy = np.meshgrid(np.append(g, S - 1), np.append(g, S - 1))
        v = np.stack((gx.ravel(), gy.ravel()), axis=-1).astype(np.float64)
        inner = (v[:, 0] > 0) & (v[:, 0] < S - 1) & (v[:, 1] > 0) & (v[:, 1] < S - 1)
        v[inner] += rng.uniform(-0.3, 0.3, (int(inner.sum()), 2)) * args.mesh_size
        meshes.append(Mesh(v, Delaunay(v).simplices.astype(np.int32), uid=k))
    images = [renderer.ResidentImage(base), renderer.ResidentImage(img1)]
    print(f'sections {S}x{S}, meshes {meshes[0].num_triangles} / {meshes[1].num_triangles} triangles')
    for rep in range(3 if args.profile else 2):
        m0, m1 = meshes[0].copy(), meshes[1].copy()
        trace = []
        prof = None
        if args.profile and rep == 2:
            import cProfile
            prof = cProfile.Profile(); prof.enable()
        t0 = time.perf_counter()
        xy0, xy1, w, strain = matcher.section_matcher(m0, m1, images[0], images[1], spacings=[280, 70], conf_thresh=0.3, residue_len=3.0, trace=trace)
        dt = time.perf_counter() - t0
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof).sort_stats('tottime').print_stats(22)
            pstats.Stats(prof).sort_stats('cumulative').print_stats(45)
            pstats.Stats(prof).print_callers('reduce')
            pstats.Stats(prof).print_callers('to_array')
            pstats.Stats(prof).print_callers('from_array')
        ex, ey = field(xy1[:, 0], xy1[:, 1])
        err = np.hypot(xy1[:, 0] - xy0[:, 0] + ex, xy1[:, 1] - xy0[:, 1] + ey)
        print(f'rep {rep}: {dt:.3f} s, rounds {[(r["blocks"], r["kept"], round(r["max_dis"], 2), r["solve"].get("iters")) for r in trace]}, '
              f'{xy0.shape[0]} matches, median error {np.median(err):.3f} px, 95 % {np.quantile(err, 0.95):.3f} px')

    if args.kernels:
        from feabas_amd import _lib
        lib, ctx = _lib.load(), _lib.ctx()
        m0, m1 = meshes[0].copy(), meshes[1].copy()
        lib.fb_prof_reset(ctx); lib.fb_prof_enable(ctx, 1)
        t0 = time.perf_counter()
        matcher.section_matcher(m0, m1, images[0], images[1], spacings=[280, 70], conf_thresh=0.3, residue_len=3.0)
        dt = time.perf_counter() - t0
        lib.fb_prof_enable(ctx, 0)
        snap = _lib.prof_snapshot()
        print(f'with the event profile on: {dt:.3f} s; kernels {sum(v[1] for v in snap.values()):.2f} ms')
        for k, v in sorted(snap.items(), key=lambda kv: -kv[1][1])[:16]:
            print(f'   {k:26s} launches {v[0]:4d}  {v[1]:8.3f} ms')

    if args.entries:
        from feabas_amd import _lib
        real = _lib.load()
        spent = {}

        class Timed:
            def __getattr__(self, name):
                f = getattr(real, name)
                if not name.startswith('fb_'):
                    return f

                def call(*a):
                    t0 = time.perf_counter()
                    try:
                        return f(*a)
                    finally:
                        e = spent.setdefault(name, [0, 0.0])
                        e[0] += 1; e[1] += time.perf_counter() - t0
                return call
        timed = Timed()
        real_load = _lib.load
        _lib.load = lambda: timed
        try:
            m0, m1 = meshes[0].copy(), meshes[1].copy()
            t0 = time.perf_counter()
            matcher.section_matcher(m0, m1, images[0], images[1], spacings=[280, 70], conf_thresh=0.3, residue_len=3.0)
            dt = time.perf_counter() - t0
        finally:
            _lib.load = real_load
        inside = sum(v[1] for v in spent.values())
        print(f'entries: wall {1e3 * dt:.1f} ms, inside the library {1e3 * inside:.1f} ms in {sum(v[0] for v in spent.values())} calls, '
              f'python between the calls {1e3 * (dt - inside):.1f} ms')
        for k, v in sorted(spent.items(), key=lambda kv: -kv[1][1])[:24]:
            print(f'   {k:34s} calls {v[0]:5d}  {1e3 * v[1]:8.2f} ms')


if __name__ == '__main__':
    main()
