"""Generate tests/golden/pp_encoder_golden.npz: the 3-level set-abstraction stack computed with the REAL reference's primitives
(/root/reference/pointnet2.py: sample_and_group, sample_and_group_all -- imported read-only with empty cv2 / torchvision stubs) and
plain torch.nn Conv2d / BatchNorm2d(eval) / ReLU / max on the grouped tensors.  The reference defines the primitives but never stacks
them, so this is the closest the reference itself gets to the encoder BASELINE.json's north_star names: every
sample, neighbour list and grouped tensor below comes out of the reference's own code.  The FPS start indices are the reference's own
torch.randint draws (pointnet2.py:66) under the seeds recorded in the file.  Run in the build container only
(`python tests/golden/make_golden_encoder.py`); the GPU box has no /root/reference and uses the committed file."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
for m in ('cv2', 'torchvision'):
    sys.modules.setdefault(m, types.ModuleType(m))
sys.path.insert(0, '/root/reference')
import pointnet2 as ref  # noqa: E402  (the reference implementation itself)

from catgrasp_amd import pointnet2 as p2  # noqa: E402  (only for the module's parameter layout / seeded weights)

torch.set_num_threads(1)
CFG = dict(channel=6, npoints=(96, 24), radii=(0.25, 0.5), nsamples=(16, 32), mlps=((32, 64, 64), (64, 128, 128), (128, 256, 512)))
SEEDS = (1234, 1235)            # torch seeds in force when level 1 / level 2 draw their FPS start


def mlp_max(new_points, convs, bns):
    h = new_points.permute(0, 3, 2, 1)                      # (B, C, K, S), the layout a Conv2d(1x1) stack consumes
    for conv, bn in zip(convs, bns):
        h = torch.relu(bn(conv(h)))
    return torch.max(h, 2)[0].permute(0, 2, 1)              # (B, S, C)


def main():
    rng = np.random.default_rng(99)
    B, N = 2, 1500
    pts = rng.uniform(-0.7, 0.7, (B, N, 3)).astype(np.float32)
    nrm = rng.normal(size=(B, N, 3)).astype(np.float32); nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
    x = np.concatenate([pts, nrm], 2)
    torch.manual_seed(5)
    enc = p2.PointNet2Encoder(**CFG)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1, generator=g); m.running_var.uniform_(0.5, 1.5, generator=g)
                m.weight.uniform_(0.5, 1.5, generator=g); m.bias.normal_(0, 0.1, generator=g)
    enc.eval()
    out = {'x': x, 'seeds': np.array(SEEDS)}
    for k, v in enc.state_dict().items():
        out['sd.' + k] = v.numpy()
    xyz, feats = torch.from_numpy(pts), torch.from_numpy(nrm)
    with torch.no_grad():
        torch.manual_seed(SEEDS[0])
        l1_xyz, g1, _, fps1 = ref.sample_and_group(CFG['npoints'][0], CFG['radii'][0], CFG['nsamples'][0], xyz, feats, returnfps=True)
        l1 = mlp_max(g1, enc.sa1.mlp_convs, enc.sa1.mlp_bns)
        torch.manual_seed(SEEDS[1])
        l2_xyz, g2, _, fps2 = ref.sample_and_group(CFG['npoints'][1], CFG['radii'][1], CFG['nsamples'][1], l1_xyz, l1, returnfps=True)
        l2 = mlp_max(g2, enc.sa2.mlp_convs, enc.sa2.mlp_bns)
        _, g3 = ref.sample_and_group_all(l2_xyz, l2)
        l3 = mlp_max(g3, enc.sa3.mlp_convs, enc.sa3.mlp_bns)[:, 0]
    out.update(fps1=fps1.numpy(), fps2=fps2.numpy(), l1_xyz=l1_xyz.numpy(), l1_points=l1.numpy(), l2_xyz=l2_xyz.numpy(), l2_points=l2.numpy(),
               global_feat=l3.numpy())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pp_encoder_golden.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), 'bytes;', 'global feature', l3.shape, float(l3.abs().max()))


if __name__ == '__main__':
    main()
