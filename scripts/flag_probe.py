"""dev probe: is the half-range flag set on the bench workload, and what does a step cost per arithmetic?"""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from catgrasp_amd import engine, synth
from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, DEFAULT_NUNOCS_CFG, GraspPredicter, NunocsPredicter
dev = torch.device('cuda:0')
gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=0), device=dev)
npred = NunocsPredicter('nut', cfg=DEFAULT_NUNOCS_CFG, state_dict=synth.make_state_dict('seg', 6, 300, seed=1), device=dev)
wl = bench.build_workload(dev, 10000, seed=0)
for prec in ('f16x3', 'bf16x3', 'f16x3'):
    engine.set_precision(prec)
    engine.half_range_violation(reset=True)
    with torch.no_grad():
        for _ in range(2): bench.run_step(wl, gp, npred)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): bench.run_step(wl, gp, npred)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(prec, f'{dt*1e3:.2f} ms/step', 'flag', engine.half_range_violation(reset=True))
    with torch.no_grad():
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): gp.score_on_device(wl['cloud_xyz'], wl['cloud_normal'], wl['ids'], wl['pose_inv'])
        torch.cuda.synchronize(); print('   scoring only', f'{(time.perf_counter() - t) / 3 * 1e3:.2f} ms', 'flag', engine.half_range_violation(reset=True))
        coords, conf, _ = npred.nocs_on_device(wl['cloud_xyz'], wl['cloud_normal'], wl['nunocs_ids'])
        torch.cuda.synchronize(); print('   after nunocs flag', engine.half_range_violation(reset=True))
