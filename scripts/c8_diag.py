"""One step of the fp8 engine (fp8 W x W contractions + fp8 copies) against the bf16 step on the same parameters: loss, network
output, the H_1 / dZ_1 / dZ_0 copies and every gradient leaf (max / rms differences, correlation).  GPU diagnostic."""
import sys

import numpy as np

sys.path.insert(0, '.')
from bayesnf_amd.engine import Engine     # noqa: E402
from tests import util                    # noqa: E402
net, model, X, y = util.make_problem(n_rows=700, width=512, depth=2)
E = 3
theta = util.random_theta(model, E, scale=0.3)
res = {}
for dt in ('fp8', 'bf16'):
  eng = Engine(net, X=X, y=y, members=E, compute_dtype=dt, pipeline='panel')
  eng.set_params(theta)
  loss, g = eng.debug_loss_and_grad()
  res[dt] = dict(loss=loss, g=g, out=eng.debug_activation(200), H1=eng.debug_activation(1), dZ0=eng.debug_activation(300), dZ1=eng.debug_activation(301))
  eng.close()
a, b = res['fp8'], res['bf16']
print('loss', a['loss'], b['loss'])
for k in ('out', 'H1', 'dZ1', 'dZ0'):
  d = np.abs(a[k] - b[k]); s = np.abs(b[k]).max()
  print(k, 'max abs diff / max', d.max() / s, 'rms diff / rms', np.sqrt((d**2).mean()) / np.sqrt((b[k]**2).mean()), 'corr', np.corrcoef(a[k].ravel(), b[k].ravel())[0, 1])
errs = util.per_leaf_rel_err(model, a['g'], b['g'])
print({k: round(v, 4) for k, v in errs.items()})
