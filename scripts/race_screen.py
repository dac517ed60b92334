#!/usr/bin/env python
"""Race screen for the ring-buffered K loops (counted vmcnt + raw s_barrier): the same forward +
backward at the benchmark size, repeated; every repetition must reproduce the first one up to
the re-ordering of f32 atomics.  A staging race shows up as an outlier tile."""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from bench import MODEL_KW, synthetic_grid     # noqa: E402
from bayesnf_amd.engine import Engine          # noqa: E402
from bayesnf_amd.spec import NetSpec           # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
X, y, scales = synthetic_grid()
worst = 0.0
for dtype, members in (('bf16', 16), ('fp8', 16), ('fp32', 4)):
  net = NetSpec(input_scales=scales, **MODEL_KW)
  eng = Engine(net, X=X, y=y, members=members, seed=1, compute_dtype=dtype)
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  loss0, g0 = eng.debug_loss_and_grad()
  scale = np.abs(g0).max(axis=1, keepdims=True)
  for r in range(reps):
    loss, g = eng.debug_loss_and_grad()
    dl = np.abs(loss - loss0).max() / np.abs(loss0).max()
    dg = (np.abs(g - g0) / scale).max()
    worst = max(worst, dl, dg)
    if dl > 1e-5 or dg > 1e-4:
      print(f'MISMATCH {dtype} rep {r}: loss {dl:.3e} grad {dg:.3e}')
      sys.exit(1)
  eng.close()
print(f'race screen ok: {reps} repetitions x 3 dtypes, worst relative deviation {worst:.2e}')
