#!/usr/bin/env python
"""Reproducibility screen of every form of the row-panel kernel added in round 3 (weight rings prefetched across
barriers, parked pre-activations, LDS feature panels, merged / split weight-gradient launches): the same loss +
gradient evaluation repeated on one engine; every repetition must reproduce the first one up to the re-ordering of
f32 atomics.  usage: python scripts/race_screen_forms.py [repetitions] [bf16|fp8]   (fp8: the forms with the W x W
contractions on the fp8 MFMA out of fp8 panel images -- round 6: one more barrier and a second dZ_L image on the L1T forms)"""
import sys

import numpy as np

sys.path.insert(0, '.')
from bayesnf_amd.engine import Engine          # noqa: E402
from tests import util                         # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
CASES = [  # width, depth, harmonics, mode, rows
    (512, 2, (2, 10), 'map', 5000), (512, 4, (2, 10), 'map', 3000), (512, 4, (2, 10), 'vi', 2000),
    (1024, 2, (2, 10), 'map', 3000), (1024, 4, (2, 10), 'map', 2000),
    (256, 2, (2, 10), 'map', 6000), (256, 3, (2, 10), 'map', 4000), (256, 2, (20, 20), 'map', 6000), (256, 4, (20, 20), 'map', 3000),
]
total, worst = 0, 0.0
for width, depth, harm, mode, rows in CASES:
  net, model, X, y = util.make_problem(n_rows=rows, width=width, depth=depth, periods=(52.1775, 365.25), harmonics=harm)
  kw = dict(mode='vi', vi_samples=3, kl_weight=0.2) if mode == 'vi' else {}
  eng = Engine(net, X=X, y=y, members=6, seed=1, compute_dtype=dtype, pipeline='panel', **kw)
  eng.init_params(0.1)
  loss0, g0 = eng.debug_loss_and_grad()
  scale = np.abs(g0).max(axis=-1, keepdims=True)
  w = 0.0
  for r in range(reps):
    loss, g = eng.debug_loss_and_grad()
    dl = np.abs(loss - loss0).max() / np.abs(loss0).max()
    dg = (np.abs(g - g0) / scale).max()
    w = max(w, dl, dg)
    if dl > 1e-5 or dg > 2e-4:
      print(f'MISMATCH W={width} depth={depth} F={net.F} {mode} rep {r}: loss {dl:.3e} grad {dg:.3e}')
      sys.exit(1)
  total += reps
  worst = max(worst, w)
  print(f'W={width} depth={depth} F={net.F} {mode} rows={rows}: {reps} repetitions, worst relative deviation {w:.2e}')
  eng.close()
print(f'ok [{dtype}]: {total} repeated evaluations over {len(CASES)} kernel forms, worst relative deviation {worst:.2e}')
