"""Alignment-scale timing of the general-mesh block matcher (matcher.bboxes_mesh_renderer_matcher): a section image of
S x S uint8 resident in HBM, a triangulated mesh of ~T triangles with a smooth non-affine field, NB blocks of B x B pixels
(the 0.7 x 400 px blocks of alignment_configs.yaml:16-23).  Prints the time of every stage and blocks / s."""
import argparse
import time

import numpy as np

from feabas_amd import _lib, matcher, renderer
from feabas_amd import constant as const
from feabas_amd.mesh import Mesh


def grid_mesh(extent, spacing, rng):
    nx, ny = int(extent / spacing) + 1, int(extent / spacing) + 1
    gx, gy = np.meshgrid(np.linspace(0, extent, nx), np.linspace(0, extent, ny))
    v = np.stack((gx.ravel(), gy.ravel()), axis=-1)
    inner = ((gx > 0) & (gx < extent) & (gy > 0) & (gy < extent)).ravel()
    v[inner] += rng.uniform(-0.25, 0.25, (int(inner.sum()), 2)) * spacing
    idx = np.arange(nx * ny).reshape(ny, nx)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate((np.stack((a, b, d), -1), np.stack((a, d, c), -1))).astype(np.int32)
    return v, tris


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=16384)
    ap.add_argument('--mesh-size', type=float, default=50.0)
    ap.add_argument('--blocks', type=int, default=1024)
    ap.add_argument('--block', type=int, default=280)
    ap.add_argument('--sigma', type=float, default=2.5)
    ap.add_argument('--tol', type=float, default=0.0)
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    S = args.size
    lib, ctx = _lib.load(), _lib.ctx()
    img = rng.integers(0, 255, (S, S), dtype=np.uint8)
    images = [renderer.ResidentImage(img), renderer.ResidentImage(np.roll(img, (3, -5), axis=(0, 1)))]
    meshes = []
    for k in range(2):
        v, tris = grid_mesh(S - 1, args.mesh_size, rng)
        s = v / S
        field = 4.0 * np.stack((np.sin(3.1 * s[:, 1] + k), np.cos(2.3 * s[:, 0] - k)), axis=-1)
        M = Mesh(v, tris)
        M.set_vertices(v + field, const.MESH_GEAR_MOVING)
        meshes.append(M)
    B = args.block
    x0 = rng.integers(0, S - B, args.blocks); y0 = rng.integers(0, S - B, args.blocks)
    bboxes = np.stack((x0, y0, x0 + B, y0 + B), axis=-1)
    print(f'image {S}x{S} u8, mesh {meshes[0].num_triangles} triangles, {args.blocks} blocks of {B}x{B}, sigma {args.sigma}, tol {args.tol}')
    rends = [renderer.MeshRenderer.from_mesh(M, image_loader=im, affine_approx_tol=args.tol) for M, im in zip(meshes, images)]
    for rep in range(args.reps):
        t0 = time.perf_counter()
        d_out, d_mask, shape, tier = rends[0].render_stack_dev(bboxes)
        t1 = time.perf_counter()
        d_f = rends[0].filter_stack_dev(d_out, d_mask, shape, args.sigma)
        t2 = time.perf_counter()
        for b in (d_out, d_mask, d_f):
            b.free()
        t3 = time.perf_counter()
        xy0, xy1, conf = matcher.bboxes_mesh_renderer_matcher(meshes[0], meshes[1], rends[0], rends[1], bboxes, bboxes, sigma=args.sigma,
                                                              affine_approx_tol=args.tol)
        t4 = time.perf_counter()
        print(f'rep {rep}: render one stack {1e3 * (t1 - t0):.1f} ms (tiers {np.bincount(tier, minlength=4)[1:]}), masked DoG {1e3 * (t2 - t1):.1f} ms, '
              f'whole matcher {1e3 * (t4 - t3):.1f} ms = {args.blocks / (t4 - t3):.0f} blocks/s, median conf {np.median(conf):.3f}')


if __name__ == '__main__':
    main()
