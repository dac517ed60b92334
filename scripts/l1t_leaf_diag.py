"""Per-leaf gradient errors of the bf16 row-panel kernel against the float64 oracle and the fp32 engine at a few
shapes (the leaves the L1T form of round 5 computes differently: Dense_L/bias and the output kernel come out of bf16
MFMAs over the LDS panel, d alpha / d gamma_L from in-lane row dots).  BNF_LIB selects the build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from bayesnf_amd.engine import Engine
from oracle import bnf_oracle as O
import util

for width, depth, n_rows in ((512, 2, 1000), (256, 2, 700), (512, 3, 600), (1024, 2, 300)):
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  theta = util.random_theta(model, 3, scale=0.3)
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  res = {}
  for name, kw in (('panel', dict(compute_dtype='bf16', pipeline='panel')), ('fp32', dict(compute_dtype='fp32'))):
    eng = Engine(net, X=X, y=y, members=3, **kw)
    eng.set_params(theta)
    res[name] = eng.debug_loss_and_grad()
    eng.close()
  e_o = util.per_leaf_rel_err(model, res['panel'][1], g_o)
  e_f = util.per_leaf_rel_err(model, res['panel'][1], res['fp32'][1])
  print(f'W={width} depth={depth} rows={n_rows}: loss rel {np.max(np.abs(res["panel"][0] / loss_o - 1)):.2e}')
  for k in e_o:
    if 'bias' in k or k.endswith(f'Dense_{depth}/kernel') or 'scale' in k or 'logit' in k:
      print(f'   {k:34s} vs oracle {e_o[k]:.2e}   vs fp32 {e_f[k]:.2e}')
