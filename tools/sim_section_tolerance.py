"""the oracle's section matcher once with exact relaxations, once with fem_ref.pcg to 1e-9, for a locked and for a floating pair:
how far a practical tolerance moves the relaxed field and the final matches (CPU only, ~10 s)"""
import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from scipy.ndimage import map_coordinates
from oracle import fem_ref, region_ref
from test_gpu_renderer import _island_pair, _texture
rng = np.random.default_rng(17)
(v0, t0, v1, t1), (M0, M1), _ = _island_pair(rng)
SH, SW = 600, 1080
base = _texture(rng, SH, SW)
yy, xx = np.meshgrid(np.arange(SH, dtype=np.float64), np.arange(SW, dtype=np.float64), indexing='ij')
ux = 3.0 * np.sin(2 * np.pi * yy / 700.0 + 0.4) + 1.0 * (xx / SW) ** 2
uy = 2.5 * np.cos(2 * np.pi * xx / 900.0) - 1.0 * (xx / SW) * (yy / SH)
img1 = np.clip(np.rint(map_coordinates(base.astype(np.float32), [yy + uy, xx + ux], order=1, mode='nearest')), 0, 255).astype(np.uint8)
kw = dict(spacings=[150, 60], sigma=2.5, conf_thresh=0.3, residue_len=3.0, min_boundary_distance=12, stiffness_lambda=0.5)
res = {}
orig = region_ref._optimize_linear
def pcg_version(tol):
    def f(meshes, links, stiffness_lambda):
        A, b, _ = fem_ref.linear_system(meshes, links, stiffness_lambda, -1.0, fem_ref.GEAR_FIXED, fem_ref.GEAR_MOVING, fem_ref.GEAR_MOVING)
        A = (0.5 * (A + A.T)).tocsr()
        d = A.diagonal(); minv = 1.0 / np.maximum(d, min(1.0, d.max() / 1000))
        x, it, rel = fem_ref.pcg(A, np.asarray(b, float), rtol=tol, maxiter=200000, minv=minv)
        print('   pcg iters', it, 'rel', rel)
        fem_ref.apply_solution(meshes, x, fem_ref.GEAR_MOVING, fem_ref.GEAR_MOVING)
    return f
for name, fn in (('exact', orig), ('pcg1e-9', pcg_version(1e-9))):
    for locked in (True, False):
        region_ref._optimize_linear = fn
        r0 = fem_ref.RefMesh(v0, t0, uid=0, locked=locked); r1 = fem_ref.RefMesh(v1, t1, uid=1)
        tr = []
        t_ = time.time()
        out = region_ref.section_match(r0, r1, base, img1, compute_strain=True, batch_size=100, trace=tr, res=None, **kw)
        print(name, 'locked' if locked else 'floating', 'matches', out[0].shape, 'strain', out[3], 'time', round(time.time()-t_,1))
        res[name, locked] = (out, tr)
for locked in (True, False):
    a, ta = res['exact', locked]; b, tb = res['pcg1e-9', locked]
    print('locked' if locked else 'floating')
    for r,(x,y) in enumerate(zip(ta, tb)):
        same = x['bboxes0'].shape == y['bboxes0'].shape and np.abs(x['bboxes0']-y['bboxes0']).max()
        print('  round', r, 'blocks', x['bboxes0'].shape[0], y['bboxes0'].shape[0], 'bbox diff', same, 'conf diff', np.abs(x['conf']-y['conf']).max() if x['conf'].shape==y['conf'].shape else None,
              'field diff', np.abs(x['field1']-y['field1']).max() if 'field1' in x else None, 'field max', np.abs(x['field1']).max() if 'field1' in x else None)
    if a[0].shape == b[0].shape:
        print('  xy0 diff', np.abs(a[0]-b[0]).max(), 'xy1 diff', np.abs(a[1]-b[1]).max(), 'w diff', np.abs(a[2]-b[2]).max(), 'strain', a[3], b[3])
