"""Scripts that drive the reference (tests/golden/make_golden.py) and the product (tests/) through the SAME statements: a walk
takes the class under test as an argument and imports neither side."""
import numpy as np
from scipy.spatial import Delaunay


def g27_mesh_walk(mesh_cls, consts, record):
    """a walk over the gears of one mesh -- the same statements drive the reference here and the product in its test; `record(tag, mesh)`
    is called after every step.  Returns the inputs."""
    rng = np.random.default_rng(27)
    gx, gy = np.meshgrid(np.arange(6) * 40.0, np.arange(5) * 35.0)
    v = np.stack((gx.ravel(), gy.ravel()), axis=-1) + rng.uniform(-6, 6, (30, 2))
    tri = Delaunay(v).simplices.astype(np.int32)
    F, M, S, I = consts.MESH_GEAR_FIXED, consts.MESH_GEAR_MOVING, consts.MESH_GEAR_STAGING, consts.MESH_GEAR_INITIAL
    m = mesh_cls(v, tri, uid=3)
    d = rng.normal(0, 2.0, v.shape); d2 = rng.normal(0, 1.0, v.shape); d3 = rng.normal(0, 1.5, v.shape)
    mask = np.zeros(30, dtype=bool); mask[[1, 4, 7, 8, 15, 22, 29]] = True
    A = np.array([[0.98, 0.05, 0.0], [-0.04, 1.03, 0.0], [5.0, -3.0, 1.0]])
    A2 = np.array([[1.01, -0.02, 0.0], [0.03, 0.97, 0.0], [-2.0, 7.5, 1.0]])
    record('s00_new', m)
    m.apply_translation((3.5, -2.0), F); record('s01_translate_fixed', m)
    m.set_field(d, gear=(F, M)); record('s02_field_fixed_to_moving', m)
    m.apply_affine(A, M); record('s03_affine_moving', m)
    m.apply_field(d2, M); record('s04_field_moving', m)
    m.apply_field(d3[mask], M, vtx_mask=mask); record('s05_masked_field_moving', m)
    m.apply_translation((1.0, 1.0), F); record('s06_translate_fixed_again', m)
    m.set_translation((4.0, -1.0), gear=(F, M)); record('s07_set_translation', m)
    m.set_affine(A2, gear=(F, M)); record('s08_set_affine', m)
    m.set_vertices(m.vertices(gear=M) + d2, S); record('s09_staging_set', m)
    m.apply_affine(A, F, vtx_mask=mask); record('s10_masked_affine_fixed', m)
    m.set_vertices(m.vertices(gear=M)[mask] + 1.25, M, vtx_mask=mask); record('s11_masked_set_moving', m)
    m.apply_translation((0.0, 0.0), M); record('s12_zero_translation', m)
    m.apply_affine(np.eye(3), M); record('s13_identity_affine', m)
    m.lock()
    m.apply_translation((9.0, 9.0), M); m.set_field(d, gear=(F, M)); m.apply_affine(A, F); m.set_vertices(v, M); record('s14_locked', m)
    m.unlock()
    m.set_field(d3, gear=(M, M)); record('s15_field_in_place', m)
    m.set_translation((2.0, 2.0), gear=(M, M)); record('s16_translation_in_place', m)
    m.set_affine(A2, gear=(F, F)); record('s17_affine_in_place', m)
    return m, v, tri, mask


def g28_cascade_walk(mesh_cls, link_cls, slm_cls, consts, g15, record):
    """SLM.optimize_affine_cascade on the six-tile system of fixture G15, every tile given a rotation + scale + offset of its own at the
    start gear so that the per-mesh fits are not translations: rigid (svd_clip (1, 1)), clipped (0.9, 1.1), free affine (None), from the
    INITIAL and from the FIXED gear, with one and with two locked tiles, and a tile no link reaches"""
    F, M, I = consts.MESH_GEAR_FIXED, consts.MESH_GEAR_MOVING, consts.MESH_GEAR_INITIAL

    def system(locked=(0,), drop_links=()):
        ms = [mesh_cls(g15['v'], g15['t'], uid=k) for k in range(6)]
        for k, m in enumerate(ms):
            th = 0.02 * (k - 2.5); sc = 1.0 + 0.01 * (k - 3)
            A = np.array([[sc * np.cos(th), sc * np.sin(th), 0.0], [-sc * np.sin(th), sc * np.cos(th) * (1.0 + 0.005 * k), 0.0], [3.0 * k - 7.0, 2.0 - 1.5 * k, 1.0]])
            m.apply_affine(A, F)
        for k in locked:
            ms[k].lock()
        links = []
        for k in range(7):
            if k in drop_links:
                continue
            a, b = (int(x) for x in g15[f'l{k}_ab'])
            links.append(link_cls(ms[a], ms[b], g15[f'l{k}_tid0'], g15[f'l{k}_tid1'], g15[f'l{k}_B0'], g15[f'l{k}_B1'], weight=g15[f'l{k}_w']))
        return ms, slm_cls(ms, links=links)
    for tag, kw, sysargs in (('rigid_f2m', dict(start_gear=F, target_gear=M, svd_clip=(1, 1)), {}),
                             ('clip_f2m', dict(start_gear=F, target_gear=M, svd_clip=(0.9, 1.1)), {}),
                             ('affine_f2m', dict(start_gear=F, target_gear=M, svd_clip=None), {}),
                             ('rigid_i2f', dict(start_gear=I, target_gear=F, svd_clip=(1, 1)), {}),
                             ('affine_two_locked', dict(start_gear=F, target_gear=M, svd_clip=None), dict(locked=(0, 5))),
                             ('unreached', dict(start_gear=F, target_gear=M, svd_clip=(1, 1)), dict(drop_links=(1, 6))),
                             ('in_place', dict(target_gear=M, svd_clip=(1, 1)), {})):
        ms, slm = system(**sysargs)
        modified = slm.optimize_affine_cascade(**kw)
        record(tag, ms, modified)


def g35_outcast_walk(mesh_cls, link_cls, slm_cls, g15, record):
    """SLM.flag_outcasts on the six tiles of G15 with different link sets and locks: one subsystem; two with a locked tile; two without any
    lock (the minority is cast out); a second call after a tile was flagged; flags kept on the meshes"""
    def system(keep, locked=()):
        ms = [mesh_cls(g15['v'], g15['t'], uid=k) for k in range(6)]
        for k in locked:
            ms[k].lock()
        links = []
        for k in keep:
            a, b = (int(x) for x in g15[f'l{k}_ab'])
            links.append(link_cls(ms[a], ms[b], g15[f'l{k}_tid0'], g15[f'l{k}_tid1'], g15[f'l{k}_B0'], g15[f'l{k}_B1'], weight=g15[f'l{k}_w']))
        return ms, slm_cls(ms, links=links)
    # links of G15: 0:(0,1) 1:(1,2) 2:(0,3) 3:(3,4) 4:(4,5) 5:(1,4) 6:(2,5)
    for tag, keep, locked in (('one', range(7), (0,)), ('two_locked', (0, 1, 4), (0,)), ('two_free', (0, 1, 4), ()), ('three_free', (0, 3), ()),
                              ('two_locks', (0, 4), (0, 5))):
        ms, slm = system(keep, locked)
        first = slm.flag_outcasts()
        record(tag, first, ms)
        second = slm.flag_outcasts()
        record(tag + '_again', second, ms)
