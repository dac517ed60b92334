"""Widths off the 64 grid (the reference accepts any width: spatiotemporal.py:217-232, models.py:263).

The engine runs such a model at the next multiple of 64 on a zero-padded COPY of the
width-dependent leaves (bnf_kernels.h k_pad_params / k_fold_grad); parameters, gradients and
optimiser state keep the reference's shapes, the fan-in scales use the true width.  Same bars
as tests/test_gpu_parity.py (fp32 engine vs float64 oracle) -- the padding must be invisible.
"""
import numpy as np
import pytest
import torch

from oracle import bnf_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _engine(net, X, y, **kw):
  from bayesnf_amd.engine import Engine
  return Engine(net, X=X, y=y, **kw)


@pytest.mark.parametrize('depth,width,n_rows,pipeline', [
    (2, 100, 300, 'layers'), (1, 7, 130, 'layers'), (3, 200, 257, 'layers'),
    (2, 100, 300, 'auto'), (1, 72, 130, 'auto'), (3, 130, 140, 'auto'), (2, 250, 200, 'auto')])
def test_forward_and_grad_fp32_any_width(depth, width, n_rows, pipeline):
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  E = 3
  theta = util.random_theta(model, E)
  for pw in (1.0, 0.0):
    eng = _engine(net, X, y, members=E, prior_weight=pw, compute_dtype='fp32', pipeline=pipeline)
    eng.set_params(theta)
    loss_d, g_d = eng.debug_loss_and_grad()
    out_o, ch = O.forward(model, theta, X, keep=True)
    loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows, prior_weight=pw)
    assert g_d.shape == (E, model.P)
    for l in range(depth - 1):   # stored hidden outputs come back at the model's width
      H = eng.debug_activation(1 + l)
      assert H.shape[-1] == width
      assert util.rel_err(H, ch['Hs'][l + 1]) < 2e-4, l
    assert util.rel_err(eng.debug_activation(200), out_o) < 2e-4
    np.testing.assert_allclose(loss_d, loss_o, rtol=2e-5)
    bad = {k: v for k, v in util.per_leaf_rel_err(model, g_d, g_o).items() if v > 5e-4}
    assert not bad, bad
    # a second evaluation gives the same gradient (the padded accumulator is cleared in between)
    loss_2, g_2 = eng.debug_loss_and_grad()
    assert util.rel_err(g_2, g_d) < 1e-5
    eng.close()


@pytest.mark.parametrize('width', [100, 40])
def test_train_full_batch_fp32_any_width(width):
  n_rows, E, steps = 200, 4, 30
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=2)
  eng = _engine(net, X, y, members=E, seed=11, learning_rate=0.005, compute_dtype='fp32')
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  theta0 = eng.get_params().astype(np.float64)
  assert theta0.shape == (E, model.P)
  mm = model.matrix_mask()
  assert np.all(np.abs(theta0[:, mm]) <= 2.0) and abs(theta0[:, mm].std() - 0.8796) < 0.04
  losses = eng.train(0, steps)
  torch.cuda.synchronize()
  theta_o, losses_o = O.train_map(model, theta0, X, y, lr=0.005, num_epochs=steps)
  np.testing.assert_allclose(losses.cpu().numpy(), losses_o, rtol=5e-5)
  assert util.rel_err(eng.get_params(), theta_o) < 2e-3
  eng.close()


def test_vi_step_any_width():
  n_rows, E, S, width = 150, 2, 3, 100
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=2)
  eng = _engine(net, X, y, mode='vi', members=E, vi_samples=S, kl_weight=0.2, seed=3,
                learning_rate=0.01, compute_dtype='fp32')
  eng.init_params(0.0)
  p0 = eng.get_params().astype(np.float64)
  mu0, rho0 = p0[0], p0[1]
  eps0 = eng.debug_vi_eps(0)
  loss_d, g_d = eng.debug_loss_and_grad(0, 0)
  loss_o, gmu_o, grho_o = O.vi_loss_and_grad(model, mu0, rho0, eps0, X, y, n_rows, 0.2)
  np.testing.assert_allclose(loss_d, loss_o * 0.2, rtol=5e-5)
  bad = {k: v for k, v in util.per_leaf_rel_err(model, g_d[0], gmu_o).items() if v > 5e-4}
  assert not bad, ('gmu', bad)
  bad = {k: v for k, v in util.per_leaf_rel_err(model, g_d[1], grho_o).items() if v > 5e-4}
  assert not bad, ('grho', bad)
  steps = 4
  eps = {s: eng.debug_vi_eps(s) for s in range(steps)}
  losses = eng.train(0, steps)
  torch.cuda.synchronize()
  mu_o, rho_o, losses_o = O.train_vi(model, mu0, rho0, X, y, lr=0.01, num_steps=steps,
                                     sample_size=S, kl_weight=0.2, eps_fn=lambda s: eps[s])
  np.testing.assert_allclose(losses.cpu().numpy(), losses_o, rtol=2e-4)
  p = eng.get_params()
  assert util.rel_err(p[0], mu_o) < 1e-3 and util.rel_err(p[1], rho_o) < 1e-3
  eng.close()


def test_forward_only_any_width():
  from bayesnf_amd.engine import Engine
  net, model, X, y = util.make_problem(n_rows=500, width=100, depth=2)
  M = 7
  theta = util.random_theta(model, M, scale=0.4)
  eng = Engine(net, members=3, forward_only=True, row_capacity=256, compute_dtype='fp32')
  loc, aux = eng.forward(torch.tensor(theta, dtype=torch.float32, device=eng.device),
                         torch.tensor(X, dtype=torch.float32, device=eng.device))
  torch.cuda.synchronize()
  mu_o, sd_o = O.predict_normal(model, theta, X)
  assert util.rel_err(loc.cpu().numpy(), mu_o) < 2e-4
  np.testing.assert_allclose(aux[:, 0].cpu().numpy(), sd_o, rtol=1e-5)
  eng.close()


@pytest.mark.parametrize('width', [500, 250])
def test_bf16_panel_pipeline_on_padded_width(width):
  """bf16, depth 2, width 500 / 250: the row-panel kernel runs at 512 / 256 on the padded copy."""
  n_rows, E = 400, 3
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=2)
  theta = util.random_theta(model, E, scale=0.3)
  res = {}
  for pipe in ('panel', 'layers'):
    eng = _engine(net, X, y, members=E, compute_dtype='bf16', pipeline=pipe)
    eng.set_params(theta)
    res[pipe] = eng.debug_loss_and_grad()
    eng.close()
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  for pipe in res:
    np.testing.assert_allclose(res[pipe][0], loss_o, rtol=5e-3)
    bad = {k: v for k, v in util.per_leaf_rel_err(model, res[pipe][1], g_o).items() if v > 6e-2}
    assert not bad, (pipe, bad)
  bad = {k: v for k, v in util.per_leaf_rel_err(model, res['panel'][1], res['layers'][1]).items() if v > 2e-2}
  assert not bad, ('panel vs layers', bad)


def test_estimator_with_reference_default_like_odd_width():
  """fit / predict through the estimator API at width 100 (rejected before this round)."""
  import pandas as pd
  from bayesnf_amd import BayesianNeuralFieldMAP
  rng = np.random.default_rng(0)
  n = 240
  df = pd.DataFrame({'t': np.tile(np.arange(60.0), 4), 'x': np.repeat(np.arange(4.0), 60)})
  df['y'] = np.sin(2 * np.pi * df.t / 12) + 0.3 * df.x + 0.1 * rng.standard_normal(n)
  m = BayesianNeuralFieldMAP(width=100, depth=2, feature_cols=['t', 'x'], target_col='y', timetype='float',
                             seasonality_periods=[12.0], fourier_degrees=[2, 2],
                             observation_model='NORMAL', standardize=['x'])
  m = m.fit(df, seed=0, ensemble_size=4, num_epochs=300, learning_rate=0.01)
  assert m.losses_.shape[-1] == 300 and np.all(np.isfinite(m.losses_))
  assert np.all(m.losses_[..., -1] < m.losses_[..., 0])
  # params_ leaves have the reference's shapes: Dense kernels (.., n_in, 100)
  shapes = [tuple(np.asarray(v).shape[2:]) for v in m.params_]
  assert (100, 100) in shapes and (100, 1) in shapes and (100,) in shapes
  means, qs = m.predict(df, quantiles=(0.5,))
  rmse = float(np.sqrt(np.mean((np.asarray(means).mean(axis=(0, 1)) - df.y.values) ** 2)))
  assert rmse < 0.5, rmse


@pytest.mark.parametrize('n_rows,width,depth,E,dtype,pipeline', [
    (1, 64, 1, 1, 'fp32', 'auto'), (2, 64, 2, 1, 'fp32', 'auto'), (3, 128, 2, 2, 'fp32', 'layers'),
    (5, 512, 2, 1, 'bf16', 'auto'), (1, 256, 2, 3, 'bf16', 'auto'), (129, 512, 2, 1, 'bf16', 'panel'),
    (257, 256, 2, 2, 'bf16', 'panel')])
def test_tiny_and_ragged_batches(n_rows, width, depth, E, dtype, pipeline):
  """Edge sizes: a single row, a single member, batches one row past a panel / tile boundary
  (padded rows must contribute exactly nothing)."""
  net, model, X, y = util.make_problem(n_rows=max(n_rows, 2), width=width, depth=depth)
  X, y = X[:n_rows], y[:n_rows]
  theta = util.random_theta(model, E, scale=0.3)
  eng = _engine(net, X, y, members=E, compute_dtype=dtype, pipeline=pipeline)
  eng.set_params(theta)
  loss, g = eng.debug_loss_and_grad()
  eng.close()
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  np.testing.assert_allclose(loss, loss_o, rtol=2e-5 if dtype == 'fp32' else 5e-3)
  bad = {k: v for k, v in util.per_leaf_rel_err(model, g, g_o).items() if v > (5e-4 if dtype == 'fp32' else 6e-2)}
  assert not bad, bad
