"""Seeded sweep over model shapes (inputs, Fourier degrees incl. 0, no seasonal block, no
interactions, depth 1-4, widths incl. non-powers of two, row counts that do not fill a tile,
every observation model, MAP / MLE) -- one forward + backward of the fp32 engine against the oracle."""
import numpy as np
import pytest

from bayesnf_amd.spec import NetSpec
from oracle import bnf_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _case(seed):
  rng = np.random.default_rng(1000 + seed)
  D = int(rng.integers(1, 5))
  degs = [int(rng.integers(0, 5)) for _ in range(D)]
  pairs = [(i, j) for i in range(D) for j in range(i + 1, D)]
  n_int = int(rng.integers(0, len(pairs) + 1)) if pairs else 0
  inter = [list(pairs[k]) for k in rng.permutation(len(pairs))[:n_int]]
  if rng.random() < 0.3:
    periods, harmonics = [], []
  else:
    periods = sorted(rng.choice([4.0, 7.0, 12.0, 52.1775, 365.25], size=int(rng.integers(1, 4)), replace=False))
    harmonics = [int(rng.integers(1, max(2, int(p // 2)) if p < 20 else 6)) for p in periods]
  kw = dict(width=int(rng.choice([64, 128, 192, 256, 320])), depth=int(rng.integers(1, 5)),
            input_scales=[float(rng.uniform(1, 300))] + [1.0] * (D - 1), fourier_degrees=degs,
            interactions=inter, seasonality_periods=[float(p) for p in periods],
            num_seasonal_harmonics=harmonics,
            observation_model=str(rng.choice(['NORMAL', 'NORMAL', 'NB', 'ZINB'])))
  n_rows = int(rng.integers(20, 400))
  X = rng.standard_normal((n_rows, D))
  X[:, 0] = rng.integers(0, 300, n_rows)
  X = X.astype(np.float32).astype(np.float64)
  if kw['observation_model'] == 'NORMAL':
    y = rng.standard_normal(n_rows) * 2 + 1
  else:
    y = rng.poisson(2.0, n_rows) * (rng.random(n_rows) > 0.3)
  y = y.astype(np.float32).astype(np.float64)
  return kw, X, y, float(rng.choice([1.0, 0.0]))


@pytest.mark.parametrize('seed', range(16))
def test_random_model_shapes_fp32(seed):
  from bayesnf_amd.engine import Engine
  kw, X, y, pw = _case(seed)
  net, model = NetSpec(**kw), O.Model(**kw)
  assert net.P == model.P and net.F == model.F
  E = 2
  theta = util.random_theta(model, E, seed=seed, scale=0.4)
  eng = Engine(net, X=X, y=y, members=E, prior_weight=pw, compute_dtype='fp32')
  eng.set_params(theta)
  loss_d, g_d = eng.debug_loss_and_grad()
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=X.shape[0], prior_weight=pw)
  np.testing.assert_allclose(loss_d, loss_o, rtol=5e-5, err_msg=str(kw))
  errs = util.per_leaf_rel_err(model, g_d, g_o)
  bad = {k: v for k, v in errs.items() if v > 1e-3}
  assert not bad, (bad, kw)
  eng.close()


@pytest.mark.parametrize('seed', range(16))
def test_random_model_shapes_bf16(seed):
  """Same sweep through the bf16 kernels (different K-tile geometry, fast activation formulas,
  fused last layer, large tiles): statistical agreement with the float64 oracle."""
  from bayesnf_amd.engine import Engine
  kw, X, y, pw = _case(seed)
  net, model = NetSpec(**kw), O.Model(**kw)
  E = 2
  theta = util.random_theta(model, E, seed=seed, scale=0.4)
  eng = Engine(net, X=X, y=y, members=E, prior_weight=pw, compute_dtype='bf16')
  eng.set_params(theta)
  loss_d, g_d = eng.debug_loss_and_grad()
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=X.shape[0], prior_weight=pw)
  # measured over these 16 shapes (scripts/sweep_bf16_diag.py, MI355X): loss 9e-6, Dense kernels
  # 3.3e-3, biases 3.0e-3, scalar leaves (scales, log_scale_adjustment, activation weight: sums of
  # O(rows x width) bf16-rounded products with cancellation) 3.0e-2.  Bars per leaf class at 2-5x that.
  np.testing.assert_allclose(loss_d, loss_o, rtol=2e-3, err_msg=str(kw))
  errs = util.per_leaf_rel_err(model, g_d, g_o)
  tol = lambda k: 1.5e-2 if k.endswith(('/kernel', '/bias')) else 6e-2
  bad = {k: v for k, v in errs.items() if v > tol(k)}
  assert not bad, (bad, kw)
  eng.close()
