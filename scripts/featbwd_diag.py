"""Per-leaf gradient error of the panel pipeline with / without the fused featurisation backward
(BNF_PANEL_FEATBWD) against the float64 oracle -- scalar feature leaves only."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from oracle import bnf_oracle as O
from tests import util
from bayesnf_amd.engine import Engine

for n_rows in (300, 1000, 4000):
  net, model, X, y = util.make_problem(n_rows=n_rows, width=512, depth=2)
  theta = util.random_theta(model, 3, scale=0.3)
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  for flag in ('0', '1'):
    os.environ['BNF_PANEL_FEATBWD'] = flag
    eng = Engine(net, X=X, y=y, members=3, compute_dtype='bf16', pipeline='panel')
    eng.set_params(theta)
    loss, g = eng.debug_loss_and_grad()
    eng.close()
    errs = util.per_leaf_rel_err(model, g, g_o)
    sel = {k: round(v, 5) for k, v in errs.items() if 'scale' in k or 'adjust' in k}
    print(n_rows, 'fused' if flag == '1' else 'kernel', sel)
