"""Measured errors of the two f32-class engines against the float64 oracle, at the shapes the fp32 parity tests use --
the numbers SURVEY 8d's fp32 gates (loss 1e-5, gradients 1e-4, parameters 1e-3 after 100 full-batch Adam steps) are held
against in tests/util.py::FP32_BARS.  For every (shape, dtype): forward output, loss, worst gradient leaf (max |g - g_o| over
the leaf's max |g_o|), and the parameters after 30 and after 100 Adam steps (max |theta - theta_o| / max |theta_o|), from
init-like parameters (engine init) and from generic ones (util.random_theta)."""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from bayesnf_amd.engine import Engine     # noqa: E402
from oracle import bnf_oracle as O        # noqa: E402
from tests import util                    # noqa: E402

SHAPES = [(2, 64, 300), (1, 128, 130), (3, 192, 257), (2, 256, 200), (2, 512, 600)]
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
  SHAPES = SHAPES[:2]

for depth, width, n_rows in SHAPES:
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  E = 3
  theta = util.random_theta(model, E)
  out_o, _ = O.forward(model, theta, X, keep=True)
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  for dt in ('fp32', 'fp32_split'):
    eng = Engine(net, X=X, y=y, members=E, compute_dtype=dt, learning_rate=0.005, seed=11)
    eng.set_params(theta)
    loss, g = eng.debug_loss_and_grad()
    out = eng.debug_activation(200)
    errs = util.per_leaf_rel_err(model, g, g_o)
    worst = max(errs, key=errs.get)
    line = (f'depth {depth} W {width:4d} N {n_rows:4d} {dt:10s}: out {util.rel_err(out, out_o):.1e} loss {np.abs(loss / loss_o - 1).max():.1e} '
            f'grad {errs[worst]:.1e} ({worst})')
    # trajectories from the engine's own initial parameters (what a fit starts from)
    eng.init_params(float(np.log(np.nanstd(y) / 2)))
    theta0 = eng.get_params().astype(np.float64)
    for steps in (30, 100):
      eng.set_params(theta0)
      l_d = eng.train(0, steps).cpu().numpy()
      torch.cuda.synchronize()
      th_o, l_o = O.train_map(model, theta0, X, y, lr=0.005, num_epochs=steps)
      line += f' | {steps} steps: losses {np.abs(l_d / l_o - 1).max():.1e} params {util.rel_err(eng.get_params(), th_o):.1e}'
      eng.close()
      eng = Engine(net, X=X, y=y, members=E, compute_dtype=dt, learning_rate=0.005, seed=11)
    eng.close()
    print(line, flush=True)
