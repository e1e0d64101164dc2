#!/usr/bin/env python
"""Wall-clock of the RANSAC hypothesis draw (2 x 10,000 x np.random.choice(8192, 4, replace=False) from numpy's stream) through
cg_host_numpy_choice_heads (AVX-512 and scalar twin), next to the full-row replay it replaced."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catgrasp_amd import transforms  # noqa: E402

for isa in (0, 1):
    ts = []
    for rep in range(3):
        np.random.seed(0); t = time.perf_counter(); transforms.NumpyHeadsDraw(8192, 4, 20000, isa=isa).result(); ts.append(time.perf_counter() - t)
    print(f'heads  isa={"avx512" if isa == 0 else "scalar"}: 20000 x 8192 in {min(ts) * 1e3:.1f} ms ({min(ts) / 20000 * 1e6:.2f} us / hypothesis)')
np.random.seed(0); t = time.perf_counter(); transforms.draw_ids_reference(8192, 4, 2000); dt = time.perf_counter() - t
print(f'full-row replay (round 3): {dt / 2000 * 1e6:.2f} us / hypothesis = {dt * 10 * 1e3:.0f} ms per 20000')
