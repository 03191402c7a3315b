"""Section-scale stitching optimisation (BASELINE config 4, FEM side): a 20 x 20 grid of tile meshes (one cartesian mesh per
tile), matches between neighbours as the matching stage delivers them, SLM.optimize_linear on the device.  Prints the
set-up / assembly / solve times and how well the injected stage errors are removed."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from feabas_amd import _lib, mesh, optimizer, constant as const
G = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T, ov, msz = 4096, 410, 150.0
rng = np.random.default_rng(0)
t0 = time.time()
nom = np.array([[gx * (T - ov), gy * (T - ov)] for gy in range(G) for gx in range(G)], dtype=np.float64)
err = rng.normal(0, 6.0, nom.shape); err[0] = 0                      # stage error of every tile (what the optimisation removes)
meshes = []
for k in range(G * G):
    m = mesh.Mesh.from_bbox((0, 0, T, T), cartesian=True, mesh_size=msz, uid=k)
    m.apply_translation(nom[k] + err[k], const.MESH_GEAR_FIXED)     # tiles placed at their (wrong) stage coordinates
    meshes.append(m)
meshes[0].lock()
slm = optimizer.SLM(meshes, stiffness_lambda=1.0, crosslink_lambda=-1.0)
print(f'{G * G} meshes, {sum(m.num_vertices for m in meshes)} nodes: {time.time() - t0:.2f} s'); t0 = time.time()
nl = 0
for gy in range(G):
    for gx in range(G):
        k = gy * G + gx
        for dx_, dy_ in ((1, 0), (0, 1)):
            if gx + dx_ >= G or gy + dy_ >= G:
                continue
            j = (gy + dy_) * G + gx + dx_
            # true geometry: tile content at nom (no error); a match = the same world point in both tiles' pixel frames
            n = 385
            if dx_:
                wx = rng.uniform(nom[j][0] + 5, nom[k][0] + T - 5, n); wy = rng.uniform(nom[k][1] + 5, nom[k][1] + T - 5, n)
            else:
                wx = rng.uniform(nom[k][0] + 5, nom[k][0] + T - 5, n); wy = rng.uniform(nom[j][1] + 5, nom[k][1] + T - 5, n)
            w = np.stack((wx, wy), -1)
            xy0 = w - nom[k] + rng.normal(0, 0.1, w.shape); xy1 = w - nom[j] + rng.normal(0, 0.1, w.shape)
            slm.add_link_from_coordinates(k, j, xy0, xy1, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_INITIAL),
                                          weight=rng.uniform(0.4, 1.0, n).astype(np.float32))
            nl += n
print(f'{len(slm.links)} links, {nl} matches: {time.time() - t0:.2f} s'); t0 = time.time()
cost = slm.optimize_linear(tol=1e-6, maxiter=None)
print(f'optimize_linear: {time.time() - t0:.2f} s, cost {cost}, PCG iterations {slm.last_solve["iters"]}')
t0 = time.time()
cost = slm.optimize_linear(tol=1e-6, maxiter=None)
print(f'second optimize_linear (pattern cached): {time.time() - t0:.2f} s, PCG iterations {slm.last_solve["iters"]}')
# residual stage error after the optimisation: tile centres relative to tile 0
c = np.array([m.vertices_w_offset(const.MESH_GEAR_MOVING).mean(axis=0) for m in meshes])
c0 = np.array([m.vertices_w_offset(const.MESH_GEAR_INITIAL).mean(axis=0) for m in meshes]) + nom
res = (c - c[0]) - (c0 - c0[0])
print(f'tile position error: before {np.abs(err).max():.2f} px, after {np.abs(res).max():.3f} px (rms {np.sqrt((res ** 2).mean()):.3f})')
