"""bench.py's CPU baseline (oracle/torch_baseline.py: the oracle's algorithm on torch CPU tensors,
members batched with torch.bmm) against the numpy oracle: same loss, same hand-derived gradient,
same Adam trajectory.  CPU only."""
import numpy as np
import torch

from oracle import bnf_oracle as O
from oracle.torch_baseline import TorchStep
from tests import util


def test_torch_baseline_matches_numpy_oracle():
  net, model, X, y = util.make_problem(n_rows=200, width=64, depth=2)
  theta = util.random_theta(model, 3, scale=0.4)
  for pw in (1.0, 0.0):
    ts = TorchStep(model, X, y, prior_weight=pw)
    with torch.no_grad():
      loss_t, g_t = ts.loss_and_grad(torch.as_tensor(theta.astype(np.float32)))
    loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=200, prior_weight=pw)
    np.testing.assert_allclose(loss_t.numpy(), loss_o, rtol=2e-5)
    bad = {k: v for k, v in util.per_leaf_rel_err(model, g_t.numpy(), g_o).items() if v > 2e-4}
    assert not bad, bad
  ts = TorchStep(model, X, y, lr=0.005)
  th_t, l_t = ts.train(theta, 8)
  th_o, l_o = O.train_map(model, theta, X, y, lr=0.005, num_epochs=8)
  np.testing.assert_allclose(l_t, l_o, rtol=1e-4)
  assert util.rel_err(th_t, th_o) < 1e-3
