import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from bayesnf_amd.engine import Engine
from bayesnf_amd.spec import NetSpec
from oracle import bnf_oracle as O
import util, test_gpu_sweep as S
cls = lambda k: 'kernel' if k.endswith('/kernel') else ('bias' if k.endswith('/bias') else 'scalar')
worst = {}; lossw = 0
for seed in range(16):
  kw, X, y, pw = S._case(seed)
  net, model = NetSpec(**kw), O.Model(**kw)
  theta = util.random_theta(model, 2, seed=seed, scale=0.4)
  eng = Engine(net, X=X, y=y, members=2, prior_weight=pw, compute_dtype='bf16')
  eng.set_params(theta)
  loss_d, g_d = eng.debug_loss_and_grad()
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=X.shape[0], prior_weight=pw)
  lossw = max(lossw, np.max(np.abs(loss_d/loss_o-1)))
  for k, v in util.per_leaf_rel_err(model, g_d, g_o).items():
    c = cls(k)
    if v > worst.get(c, (0,))[0]: worst[c] = (v, k, seed, kw['width'], kw['depth'])
  eng.close()
print('loss', lossw); print(worst)
