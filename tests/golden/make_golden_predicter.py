"""Generate tests/golden/predicter_golden.npz by running the REAL `predicter.GraspPredicter.predict_batch`
(predicter.py:67-94) and the network + decode part of `NunocsPredicter.predict` (predicter.py:135-150) on the CPU.
Build container only.  Uninstallable imports are inert stubs; `Tensor.cuda()` / `torch.cuda.empty_cache()` are made no-ops
so the reference's own code path (python transform loop with numpy-global-RNG resampling -> chunks of 200 -> the real
pointnet2.PointNetCls / PointNetSeg -> softmax / argmax decode) runs unmodified on CPU tensors."""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


class StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    roots = ('cv2', 'torchvision', 'open3d', 'trimesh', 'autolab_core', 'pybullet', 'pybullet_data', 'mayavi', 'pybullet_tools', 'pyrender',
             'imgaug', 'skimage', 'ikfast_pybind', 'my_cpp', 'data_reader', 'renderer', 'PointGroup', 'spconv', 'torchprof')

    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name; m.__path__ = []; m.__spec__ = spec; m.__all__ = []
        return m

    def exec_module(self, module):
        pass


tf_mod = types.ModuleType('transformations'); tf_mod.__all__ = []
sys.modules['transformations'] = tf_mod
sys.meta_path.insert(0, StubFinder())
sys.path.insert(0, '/root/reference')
import dataset_grasp  # noqa: E402
import dataset_nunocs  # noqa: E402
import pointnet2 as ref_pn  # noqa: E402
import predicter as ref_pred  # noqa: E402

from catgrasp_amd import synth  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self          # run the reference's .cuda() path on the CPU
torch.cuda.empty_cache = lambda: None
torch.set_num_threads(1)

rng = np.random.default_rng(21)
ob = synth.make_scene(1, 2300, 13)[0]
poses = synth.make_candidates(ob, 7, rng)
mean = rng.normal(0, 0.002, 6); std = rng.uniform(0.004, 0.3, 6)
out = {'xyz': ob['xyz'], 'normal': ob['normal'], 'poses': poses, 'mean': mean, 'std': std}

# ---- GraspPredicter.predict_batch ----
cfg = {'n_pts': 2048, 'input_channel': 6, 'classes': [0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.01], 'mean': mean, 'std': std}
gp = types.SimpleNamespace(cfg=cfg)
gp.dataset = types.SimpleNamespace(cfg=cfg, phase='test')
gp.dataset.transform = lambda data, pose: dataset_grasp.GraspDataset.transform(gp.dataset, data, pose)
model = ref_pn.PointNetCls(6, 10)
model.load_state_dict(synth.make_state_dict('cls', 6, 10, seed=77))
gp.model = model.eval()
np.random.seed(123)
ret = ref_pred.GraspPredicter.predict_batch(gp, {'cloud_xyz': ob['xyz'].copy(), 'cloud_normal': ob['normal'].copy()}, list(poses))
out['grasp_labels'] = np.array([r[0] for r in ret]); out['grasp_conf'] = np.array([r[1] for r in ret]); out['grasp_probs'] = np.array([r[2] for r in ret])

# ---- NunocsPredicter.predict: the lines before the RANSAC (predicter.py:135-150), executed verbatim ----
ncfg = {'n_pts': 8192, 'input_channel': 6, 'ce_loss_bins': 100}
seg = ref_pn.PointNetSeg(6, 300)
seg.load_state_dict(synth.make_state_dict('seg', 6, 300, seed=78))
seg.eval()
nds = types.SimpleNamespace(cfg=ncfg, phase='test')
np.random.seed(321)
with torch.no_grad():
    data = {'cloud_xyz': ob['xyz'].copy(), 'cloud_normal': ob['normal'].copy()}
    data['cloud_nocs'] = np.zeros(data['cloud_xyz'].shape)
    data['cloud_rgb'] = np.zeros(data['cloud_xyz'].shape)
    data_transformed = dataset_nunocs.NunocsIsolatedDataset.transform(nds, data)
    input_data = torch.from_numpy(data_transformed['input']).cuda().float().unsqueeze(0)
    pred = seg(input_data)[0].reshape(-1, 3, ncfg['ce_loss_bins'])
    bin_resolution = 1 / ncfg['ce_loss_bins']
    pred_coords = pred.argmax(dim=-1).float() * bin_resolution
    probs = pred.softmax(dim=-1)
    confidence_z = torch.gather(probs[:, 2, :], dim=-1, index=pred[:, 2, :].argmax(dim=-1).unsqueeze(-1)).data.cpu().numpy().reshape(-1)
    nocs_cloud = pred_coords.data.cpu().numpy() - 0.5
out['nocs_cloud'] = nocs_cloud; out['nocs_conf_z'] = confidence_z; out['nocs_keep_ids'] = data_transformed['keep_ids']
srt = np.sort(pred.numpy(), axis=-1)
out['nocs_top2_gap'] = (srt[..., -1] - srt[..., -2]).astype(np.float32)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'predicter_golden.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path), 'bytes')

# ---- BASELINE.json configs[0] (C1) at its stated size: one 'nut' instance, 2048-pt cloud, 256 grasp candidates through the REAL
# ---- GraspPredicter.predict_batch on the CPU (n_valid == n_pts: every pose draws a full permutation of the cloud) ----
torch.set_num_threads(min(8, os.cpu_count() or 1))
rng1 = np.random.default_rng(31)
ob1 = synth.make_scene(1, 2048, 17)[0]
poses1 = synth.make_candidates(ob1, 256, rng1)
cfg1 = {'n_pts': 2048, 'input_channel': 6, 'classes': [0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.01]}      # no normaliser (predicter.py:48-56 optional)
gp1 = types.SimpleNamespace(cfg=cfg1)
gp1.dataset = types.SimpleNamespace(cfg=cfg1, phase='test')
gp1.dataset.transform = lambda data, pose: dataset_grasp.GraspDataset.transform(gp1.dataset, data, pose)
model1 = ref_pn.PointNetCls(6, 10)
model1.load_state_dict(synth.make_state_dict('cls', 6, 10, seed=79))
gp1.model = model1.eval()
np.random.seed(456)
ret1 = ref_pred.GraspPredicter.predict_batch(gp1, {'cloud_xyz': ob1['xyz'].copy(), 'cloud_normal': ob1['normal'].copy()}, list(poses1))
after = np.random.randint(0, 2 ** 31, 4)          # numpy's global generator as the reference's loop leaves it
path1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'predicter_golden_c1.npz')
np.savez_compressed(path1, xyz=ob1['xyz'], normal=ob1['normal'], poses=poses1, grasp_labels=np.array([r[0] for r in ret1]),
                    grasp_conf=np.array([r[1] for r in ret1]), grasp_probs=np.array([r[2] for r in ret1]).astype(np.float32), rng_after=after)
print('wrote', path1, os.path.getsize(path1), 'bytes')
