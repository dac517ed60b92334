"""Repeat one gradient evaluation many times and report every evaluation that deviates from the
first by more than atomics-order noise -- hunting a rare (~1 in 50 test runs) wrong-gradient event."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from tests import util
from bayesnf_amd.engine import Engine

def hunt(name, n_rep, fresh_engine, **kw):
  net, model, X, y = util.make_problem(n_rows=kw.pop('n_rows'), width=kw.pop('width'), depth=kw.pop('depth'))
  E = kw.pop('members', 3)
  theta = util.random_theta(model, E, scale=kw.pop('scale', 0.5))
  ref = None; bad = 0; eng = None
  for it in range(n_rep):
    if eng is None or fresh_engine:
      if eng is not None: eng.close()
      eng = Engine(net, X=X, y=y, members=E, **kw)
      eng.set_params(theta)
    loss, g = eng.debug_loss_and_grad()
    if ref is None: ref = (loss.copy(), g.copy()); continue
    errs = util.per_leaf_rel_err(model, g, ref[1])
    worst = max(errs.values())
    if worst > 2e-3 or np.abs(loss / ref[0] - 1).max() > 1e-4:
      bad += 1
      print(name, 'iteration', it, 'DEVIATES: loss', np.abs(loss / ref[0] - 1).max(), {k: round(v, 5) for k, v in errs.items() if v > 2e-3}, flush=True)
  eng.close()
  print(name, 'done', n_rep, 'evaluations,', bad, 'deviating', flush=True)

if __name__ == '__main__':
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
  hunt('fp32 layers 3x192 fresh engines', n, True, n_rows=257, width=192, depth=3, compute_dtype='fp32', pipeline='layers', prior_weight=1.0)
  hunt('fp32 layers 3x192 one engine', n, False, n_rows=257, width=192, depth=3, compute_dtype='fp32', pipeline='layers', prior_weight=1.0)
  hunt('bf16 panel 2x512', n, False, n_rows=2000, width=512, depth=2, compute_dtype='bf16', pipeline='panel', scale=0.3)
  hunt('fp32 auto 2x512', n // 2, False, n_rows=2000, width=512, depth=2, compute_dtype='fp32', scale=0.3)
