"""the drop-in modules under nn.DataParallel, as the reference's trainers wrap them (trainer_grasp.py:33,
trainer_nunocs.py:35: `self.model = nn.DataParallel(self.model)`; checkpoints are then saved with 'module.'-prefixed keys,
trainer_grasp.py:66-70) -- train mode (torch ops, backward, optimiser step) and eval mode (the HIP kernels) through the wrapper,
and a 'module.'-prefixed legacy checkpoint loaded through it."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from catgrasp_amd import synth
from oracle import pointnet_ref as oref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('kind,n_out,N', [('cls', 10, 512), ('seg', 30, 320)])
def test_dataparallel_train_eval_and_module_prefixed_checkpoint(cuda_device, tmp_path, kind, n_out, N):
    from catgrasp_amd import pointnet2 as p2
    from catgrasp_amd import predicter
    cls = p2.PointNetCls if kind == 'cls' else p2.PointNetSeg
    sd = synth.make_state_dict(kind, 6, n_out, seed=21)
    # the trainer's side: wrap, train a step, save the wrapper's state_dict in the legacy (non-zip) format
    model = nn.DataParallel(cls(6, n_out)).to(cuda_device)                    # trainer_grasp.py:33
    model.load_state_dict({'module.' + k: v for k, v in sd.items()})
    assert all(k.startswith('module.') for k in model.state_dict())
    path = str(tmp_path / 'best_val.pth.tar')
    torch.save({'epoch': 1, 'state_dict': model.state_dict(), 'best_res': 0.0}, path, _use_new_zipfile_serialization=False)
    # the predicter's side: load_model semantics (Utils.py:135-148) strip the prefix ...
    plain = cls(6, n_out).to(cuda_device)
    plain.load_state_dict(predicter.load_state_dict(path))
    # ... and the prefixed checkpoint loads through a wrapper as it is
    model2 = nn.DataParallel(cls(6, n_out)).to(cuda_device)
    model2.load_state_dict(torch.load(path, map_location='cpu', weights_only=False)['state_dict'])
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4, N, 6, generator=g) * 0.5).to(cuda_device)
    # eval through the wrapper: the HIP kernels, same bits as the bare module, the oracle within 1e-4
    model2.eval(); plain.eval()
    with torch.no_grad():
        y_dp, tf_dp = model2(x)
        y, tf = plain(x)
    assert torch.equal(y_dp, y) and torch.equal(tf_dp, tf)
    fwd = oref.pointnet_cls_forward if kind == 'cls' else oref.pointnet_seg_forward
    ref = fwd(sd, x.cpu())[0]
    assert (y_dp.cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    # train through the wrapper: torch ops, backward reaches the first layer, the optimiser moves the weights (trainer_grasp.py:45-63)
    model2.train()
    opt = torch.optim.SGD(model2.parameters(), lr=0.05)
    out, _ = model2(x)
    if kind == 'cls':
        loss = F.cross_entropy(out, torch.tensor([0, 3, 7, 9], device=cuda_device))
    else:
        loss = F.cross_entropy(out.reshape(-1, n_out), torch.arange(4 * N, device=cuda_device) % n_out)
    opt.zero_grad(); loss.backward()
    assert float(model2.module.feat.stn.conv1.weight.grad.abs().sum()) > 0
    opt.step()
    # eval again: the HIP path sees the updated parameters and running statistics (cache keyed on tensor versions)
    model2.eval()
    with torch.no_grad():
        y2, _ = model2(x)
    assert (y2 - y_dp).abs().max().item() > 1e-4
    sd2 = {k.replace('module.', ''): v.detach().cpu() for k, v in model2.state_dict().items()}
    ref2 = fwd(sd2, x.cpu())[0]
    assert (y2.cpu() - ref2).abs().max().item() <= 1e-4 * max(1.0, ref2.abs().max().item())
    # the set-abstraction encoder wraps the same way
    enc = nn.DataParallel(p2.PointNet2Encoder(channel=6, npoints=(64, 16), radii=(0.4, 0.8), nsamples=(16, 16))).to(cuda_device).eval()
    with torch.no_grad():
        gfeat, _ = enc(x, start=(torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long)))
    assert gfeat.shape == (4, 1024) and bool(torch.isfinite(gfeat).all())
