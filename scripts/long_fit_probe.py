"""A LONG fit at the full C2 size (N = 10,232, F = 57, W = 512, depth 2, 16 members; bench.py's synthetic grid) under every
arithmetic from identical initial parameters: final loss, RMSE of the ensemble-mean prediction on the training rows, finiteness --
what the short statistical gates of tests/test_gpu_fp8.py / test_gpu_fullsize.py extrapolate to (the reference's default MAP fit is
5,000 epochs: spatiotemporal.py:480-489)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import bench                                  # noqa: E402
from bayesnf_amd.engine import Engine         # noqa: E402
from bayesnf_amd.spec import NetSpec          # noqa: E402
from oracle import bnf_oracle as O            # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
X, y, input_scales = bench.synthetic_grid()
net = NetSpec(input_scales=input_scales, **bench.MODEL_KW)
model = O.Model(input_scales=input_scales, **bench.MODEL_KW)
E, theta0, ref = 16, None, None
for dt in ('fp32_split', 'bf16', 'fp8'):
  eng = Engine(net, X=X, y=y, members=E, seed=7, learning_rate=0.005, compute_dtype=dt)
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  if theta0 is None:
    theta0 = eng.get_params()
  else:
    eng.set_params(theta0)
  t0 = time.time()
  losses = eng.train(0, steps).cpu().numpy()
  torch.cuda.synchronize()
  dt_s = time.time() - t0
  th = eng.get_params().astype(np.float64)
  eng.close()
  pred = np.asarray(O.forward(model, th, X[:4096])).mean(axis=0)
  rmse = float(np.sqrt(np.mean((pred - y[:4096]) ** 2)))
  fl = float(losses[:, -1].mean())
  ref = ref or (fl, rmse)
  print(f'{dt:10s} {steps} steps in {dt_s:6.1f} s: finite {bool(np.all(np.isfinite(losses)) and np.all(np.isfinite(th)))}  final loss {fl:.2f} '
        f'({fl / ref[0] - 1:+.5f})  rmse of the ensemble mean (first 4096 rows) {rmse:.5f} ({rmse / ref[1] - 1:+.4f})  '
        f'noise scale {float(np.mean(0.01 + np.exp(th[:, model.leaf["log_noise_scale"].offset]))):.4f}', flush=True)
