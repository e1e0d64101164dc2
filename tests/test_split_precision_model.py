"""CPU model of the split-precision arithmetic (scripts/split_precision_sim.py emulates the 16-bit piece rounding of the HIP
kernels inside the float32 oracle network): the error classes DESIGN 4.1b quotes -- half pieces at float32's own distance from the
float64 evaluation, bf16 pieces about ten times further, both far inside the 1e-4 bar -- hold on a small instance."""
import importlib.util
import os

import numpy as np
import torch

from catgrasp_amd import synth
from oracle import pointnet_ref as oref

_spec = importlib.util.spec_from_file_location('split_precision_sim', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                 'scripts', 'split_precision_sim.py'))
sim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(sim)


def test_error_classes_of_the_split_schemes():
    rng = np.random.default_rng(2)
    sd = synth.make_state_dict('cls', 6, 10, seed=12, gain=1.6)
    x = torch.from_numpy(rng.normal(0, 0.5, (3, 512, 6)).astype(np.float32))
    y64, _ = oref.pointnet_cls_forward(sd, x, torch.float64)
    y32, _ = oref.pointnet_cls_forward(sd, x)
    e32 = float(((y32 - y64).abs() / y64.abs().clamp(min=1)).max())
    e_half = sim.logits_error('f16x3', sd, x, y64)
    e_bf = sim.logits_error('bf16x3', sd, x, y64)
    e_fp8 = sim.logits_error('f16+2xfp8', sd, x, y64)
    e_two = sim.logits_error('f16+1xf16', sd, x, y64)
    print(f'f32 {e32:.1e}  f16x3 {e_half:.1e}  bf16x3 {e_bf:.1e}  f16+2xfp8 {e_fp8:.1e}  f16+1xf16 {e_two:.1e}')
    assert e_half <= max(4 * e32, 5e-6)             # half pieces: float32 class
    assert e_half < e_bf <= 1e-4 / 2                 # bf16 pieces: coarser, still well inside the bar
    assert e_fp8 > 2 * e_half and e_two > 1e-4       # the cheaper schemes that were rejected
