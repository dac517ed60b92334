"""fp32 engine against the float64 oracle on one step and 30 Adam steps (C2-like shape, depth 2 and 3): loss and
per-leaf gradient errors -- run once with the default build and once with BNF_LIB pointing at a -DBNF_FP32_FAST=1 build."""
import sys

import numpy as np

sys.path.insert(0, '.')
from bayesnf_amd.engine import Engine     # noqa: E402
from oracle import bnf_oracle as O        # noqa: E402
from tests import util                    # noqa: E402

for width, depth, n_rows in ((512, 2, 1000), (192, 3, 257)):
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  E = 4
  theta = util.random_theta(model, E, scale=0.3)
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  eng = Engine(net, X=X, y=y, members=E, compute_dtype='fp32', learning_rate=0.01)
  eng.set_params(theta)
  loss, g = eng.debug_loss_and_grad()
  errs = util.per_leaf_rel_err(model, g, g_o)
  worst = max(errs, key=errs.get)
  th_o, l_o = O.train_map(model, theta, X, y, lr=0.01, num_epochs=30)
  l_d = eng.train(0, 30).cpu().numpy()
  print(f'W={width} depth={depth}: loss rel err {np.abs(loss / loss_o - 1).max():.2e}; worst gradient leaf {worst} {errs[worst]:.2e}; '
        f'30 Adam steps: loss path rel err {np.abs(l_d / l_o - 1).max():.2e}, parameters {util.rel_err(eng.get_params(), th_o):.2e}')
  eng.close()
