"""Pins the oracle's hand-derived backward (oracle/bnf_oracle.py) with
(i) fp64 central finite differences and (ii) torch-CPU autograd on an
independently written forward.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import bnf_oracle as O
from tests import util


def small_model(obs='NORMAL', depth=2, interactions=((0, 1), (1, 2))):
  return O.Model(width=16, depth=depth, input_scales=[40.0, 1.0, 1.0],
                 fourier_degrees=[3, 2, 0], interactions=list(interactions),
                 seasonality_periods=[4.0, 12.5], num_seasonal_harmonics=[2, 3],
                 observation_model=obs)


def random_problem(model, E=3, B=37, seed=0, counts=False):
  rng = np.random.default_rng(seed)
  theta = 0.4 * rng.standard_normal((E, model.P))
  X = np.stack([rng.integers(0, 40, B).astype(float), rng.standard_normal(B),
                rng.standard_normal(B)], axis=1)
  y = rng.poisson(3.0, B).astype(float) if counts else rng.standard_normal(B) * 2 + 1
  return theta, X, y


@pytest.mark.parametrize('obs', ['NORMAL', 'NB', 'ZINB'])
@pytest.mark.parametrize('pw', [1.0, 0.0])
def test_grad_matches_finite_differences(obs, pw):
  model = small_model(obs)
  theta, X, y = random_problem(model, counts=obs != 'NORMAL')
  loss, g = O.map_loss_and_grad(model, theta, X, y, n_total=100, prior_weight=pw,
                                f32_trig_args=False)
  loss2 = O.map_loss(model, theta, X, y, 100, pw, f32_trig_args=False)
  np.testing.assert_allclose(loss, loss2, rtol=1e-12)
  rng = np.random.default_rng(1)
  idx = rng.choice(model.P, size=60, replace=False)
  # make sure every scalar leaf is probed
  idx = np.unique(np.concatenate([idx, [lf.offset for lf in model.leaves]]))
  h = 1e-6
  for p in idx:
    tp, tm = theta.copy(), theta.copy()
    tp[:, p] += h
    tm[:, p] -= h
    fd = (O.map_loss(model, tp, X, y, 100, pw, f32_trig_args=False) -
          O.map_loss(model, tm, X, y, 100, pw, f32_trig_args=False)) / (2 * h)
    np.testing.assert_allclose(g[:, p], fd, rtol=2e-5, atol=2e-6,
                               err_msg=f'param {p}')


def torch_forward(model, th, X):
  """Independent restatement of models.py:212-273 with torch ops (one member)."""
  def leaf(name):
    lf = model.leaf[name]
    return th[lf.offset:lf.offset + lf.size].reshape(lf.shape)
  sp = torch.nn.functional.softplus
  x = torch.as_tensor(X)
  s = torch.as_tensor(model.input_scales) * torch.exp(leaf('log_scale_adjustment'))
  u = x / s
  feats = [u]
  for d, deg in enumerate(model.fourier_degrees):
    if deg > 0:
      k = torch.arange(deg, dtype=torch.float64)
      yk = 2 * np.pi * 2.0**k * u[:, d:d + 1]
      feats.append(torch.cat([torch.cos(yk), torch.sin(yk)], 1) / torch.cat([k + 1, k + 1]))
  f = torch.as_tensor(model.freqs.astype(np.float64))
  hh = torch.as_tensor(model.harm.astype(np.float64))
  ys = 2 * np.pi * f * x[:, 0:1]
  feats.append(torch.cat([torch.cos(ys), torch.sin(ys)], 1) / torch.cat([hh, hh]))
  if len(model.interactions):
    feats.append(torch.stack([u[:, p] * u[:, q] for p, q in model.interactions], 1))
  names = [g[4] for g in model.groups]
  feats = [f_ for f_ in feats if f_.shape[1] > 0]
  h = torch.cat([f_ * sp(leaf(n)) for f_, n in zip(feats, names)], 1)
  a_w = torch.sigmoid(leaf('logit_activation_weight'))
  for l in range(model.depth):
    h = h / np.sqrt(h.shape[1])
    z = sp(leaf(f'inv_sp_layer_scale{l}')) * (h @ leaf(f'Dense_{l}/kernel') + leaf(f'Dense_{l}/bias'))
    h = a_w * torch.nn.functional.elu(z) + (1 - a_w) * torch.tanh(z)
  h = h / np.sqrt(h.shape[1])
  L = model.depth
  return sp(leaf('inv_sp_output_scale')) * (h @ leaf(f'Dense_{L}/kernel') + leaf(f'Dense_{L}/bias'))[:, 0]


@pytest.mark.parametrize('depth', [1, 3])
def test_grad_matches_torch_autograd(depth):
  model = small_model(depth=depth)
  theta, X, y = random_problem(model, E=2, B=29, seed=3)
  loss, g = O.map_loss_and_grad(model, theta, X, y, n_total=77, prior_weight=1.0,
                                f32_trig_args=False)
  out_o = O.forward(model, theta, X, f32_trig_args=False)
  for e in range(theta.shape[0]):
    th = torch.tensor(theta[e], dtype=torch.float64, requires_grad=True)
    out = torch_forward(model, th, X)
    np.testing.assert_allclose(out.detach().numpy(), out_o[e], rtol=1e-10, atol=1e-12)
    sigma = 0.01 + torch.exp(th[model.leaf['log_noise_scale'].offset])
    ll = torch.distributions.Normal(out, sigma).log_prob(torch.as_tensor(y)).sum()
    loc = torch.as_tensor(model.prior_loc())
    z = th - loc
    lp = (-z - 2 * torch.nn.functional.softplus(-z)).sum()
    lt = -(ll * (77 / len(y)) + lp)
    lt.backward()
    np.testing.assert_allclose(loss[e], lt.item(), rtol=1e-11)
    np.testing.assert_allclose(g[e], th.grad.numpy(), rtol=1e-8, atol=1e-10)


def test_vi_grad_matches_torch_autograd():
  model = small_model(depth=2)
  rng = np.random.default_rng(5)
  E, S, B = 2, 3, 23
  _, X, y = random_problem(model, E=E, B=B, seed=7)
  mu = 0.3 * rng.standard_normal((E, model.P))
  rho = -1.0 + 0.2 * rng.standard_normal((E, model.P))
  eps = rng.standard_normal((E, S, model.P))
  kl = 0.2
  loss, gmu, grho = O.vi_loss_and_grad(model, mu, rho, eps, X, y, n_total=60, kl_weight=kl,
                                       f32_trig_args=False)
  for e in range(E):
    m = torch.tensor(mu[e], requires_grad=True)
    r = torch.tensor(rho[e], requires_grad=True)
    sig = 1e-4 + torch.nn.functional.softplus(r)
    total = 0.0
    for s in range(S):
      z = m + sig * torch.as_tensor(eps[e, s])
      out = torch_forward(model, z, X)
      sigma = 0.01 + torch.exp(z[model.leaf['log_noise_scale'].offset])
      ll = torch.distributions.Normal(out, sigma).log_prob(torch.as_tensor(y)).sum()
      zz = z - torch.as_tensor(model.prior_loc())
      lp = (-zz - 2 * torch.nn.functional.softplus(-zz)).sum()
      logq = torch.distributions.Normal(m, sig).log_prob(z).sum()
      total = total + (logq - lp - ll * (60 / B) / kl) / S
    total.backward()
    np.testing.assert_allclose(loss[e], total.item(), rtol=1e-10)
    np.testing.assert_allclose(gmu[e], m.grad.numpy(), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(grho[e], r.grad.numpy(), rtol=1e-7, atol=1e-9)


def test_adam_matches_torch_adam():
  rng = np.random.default_rng(0)
  theta = rng.standard_normal((2, 50))
  p = torch.tensor(theta.copy(), requires_grad=True)
  opt = torch.optim.Adam([p], lr=0.005, betas=(0.9, 0.999), eps=1e-8)
  m = np.zeros_like(theta)
  v = np.zeros_like(theta)
  for t in range(1, 8):
    g = rng.standard_normal(theta.shape)
    theta, m, v = O.adam_update(theta, m, v, g, t, 0.005)
    p.grad = torch.tensor(g)
    opt.step()
  # torch: eps added to sqrt(v)/sqrt(bc2); optax: to sqrt(v/bc2) -- identical algebra
  np.testing.assert_allclose(theta, p.detach().numpy(), rtol=1e-9, atol=1e-12)


def test_chandrupatla_quantiles():
  rng = np.random.default_rng(2)
  means = rng.standard_normal((2, 4, 50)) * 3
  scales = 0.5 + rng.random((2, 4))
  for q in (0.025, 0.5, 0.975):
    x = O.normal_quantile_via_root(means, scales, q)
    np.testing.assert_allclose(O.mixture_cdf(means, scales, x), q, atol=1.1e-5)
  x = O.approximate_normal_quantile(means, scales, 0.5)
  np.testing.assert_allclose(x, means.reshape(-1, 50).mean(0), atol=1e-12)


@pytest.mark.parametrize('obs', ['NB', 'ZINB'])
def test_count_quantiles_against_pmf_summation(obs):
  """count_forecast / count_quantile_via_root vs a brute-force cumulative sum of the pmf
  (nb_log_prob / zinb_log_prob) and sampled moments."""
  _, model, X, _ = util.make_problem(n_rows=40, width=16, depth=1, observation_model=obs)
  theta = util.random_theta(model, 5, scale=0.4)
  out = O.forward(model, theta, X)
  fc = O.count_forecast(model, theta, out)
  ks = np.arange(0, 4000, dtype=np.float64)
  tc, logits = fc['tc'][:, 0], fc['logits']
  lp = np.stack([(O.zinb_log_prob(np.full_like(logits, k), tc, logits, fc['pi']) if obs == 'ZINB'
                  else O.nb_log_prob(np.full_like(logits, k), tc, logits)) for k in ks])
  pmf = np.exp(lp)                                       # (K, E, N)
  keep = pmf.sum(axis=0) > 1 - 1e-9                      # rows whose mass fits in [0, K)
  assert keep.mean() > 0.5
  np.testing.assert_allclose((pmf * ks[:, None, None]).sum(axis=0)[keep], fc['mean'][keep], rtol=1e-6)
  cdf = np.cumsum(pmf, axis=0)
  np.testing.assert_allclose(cdf[3][keep], O.count_cdf(fc, np.full((1, 1), 3.0))[keep], rtol=1e-9)
  rows = keep.all(axis=0)
  mix = cdf.mean(axis=1)                                 # (K, N)
  for q in (0.1, 0.5, 0.9):
    brute = np.argmax(mix >= q, axis=0).astype(np.float64)
    got = O.count_quantile_via_root(fc, q)
    assert np.mean(got[rows] == brute[rows]) > 0.95
    assert np.all(np.abs(got[rows] - brute[rows]) <= 1)
