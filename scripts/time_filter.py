import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from catgrasp_amd import my_cpp, synth
objs = synth.make_scene(8, 2500, 0); g = synth.make_gripper(); bg = synth.background_points(objs, 0, g['diameter'])
P = torch.from_numpy(synth.make_candidates(objs[0], 5000, np.random.default_rng(1)).astype(np.float32).reshape(-1, 16)).cuda()
sym = torch.eye(4).reshape(1, 16).cuda(); I4 = np.eye(4, dtype=np.float32)
def tess(V, F, n):
    for _ in range(n):
        nv = len(V); newV = [V]; newF = []
        for f in F:
            a, b, c = V[f[0]], V[f[1]], V[f[2]]
            newV.append(np.stack([(a + b) / 2, (b + c) / 2, (c + a) / 2]).astype(np.float32)); i0 = nv; nv += 3
            newF += [[f[0], i0, i0 + 2], [i0, f[1], i0 + 1], [i0 + 2, i0 + 1, f[2]], [i0, i0 + 1, i0 + 2]]
        V = np.concatenate(newV).astype(np.float32); F = np.array(newF, dtype=np.int32)
    return V, F
for n in (0, 2, 4):
    V, F = tess(g['vertices'], g['faces'], n); Ve, Fe = tess(g['enclosed_vertices'], g['enclosed_faces'], n)
    for accel in (False, True):
        sc = my_cpp.GripperScene(V, F, Ve, Fe, objs[0]['xyz'], bg, 0.0005, accel=accel)
        for _ in range(2):
            c, _, _ = my_cpp.filter_on_device(sc, P, sym, I4, I4, I4, I4, g['gripper_in_grasp'], True, False, False)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(3):
            c, _, _ = my_cpp.filter_on_device(sc, P, sym, I4, I4, I4, I4, g['gripper_in_grasp'], True, False, False)
        torch.cuda.synchronize(); dt = (time.time() - t) / 3
        print(f'tris={len(F)} accel={accel}: {dt*1e3:.2f} ms per 5000 poses ({5000/dt:.0f} poses/s) keep={(c==0).sum().item()}', flush=True)
