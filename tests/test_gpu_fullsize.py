"""Size-independent properties at BASELINE.json's full configuration sizes (C2, and C3 / C4-shaped
cases scaled to a few members), where the oracle is too slow: member-shard invariance, loss
decrease, fp32-vs-bf16 agreement, finite VI / minibatch training at depth 4 and widths 768 / 1024."""
import numpy as np
import pytest
import torch

from bayesnf_amd.spec import NetSpec

pytestmark = pytest.mark.gpu


def _grid(T=522, S=20, seed=1234):
  """The bench's C2 grid (bench.py:synthetic_grid), regenerated here so the test is self-contained."""
  rng = np.random.default_rng(seed)
  lat, lon = rng.uniform(-1, 1, S), rng.uniform(-1, 1, S)
  lat, lon = (lat - lat.mean()) / lat.std(), (lon - lon.mean()) / lon.std()
  tt, ss = np.meshgrid(np.arange(T, dtype=np.float64), np.arange(S), indexing='ij')
  tt, ss = tt.ravel(), ss.ravel()
  keep = ~((ss < 4) & (tt >= T - 52))
  t, s = tt[keep], ss[keep]
  X = np.stack([t, lat[s], lon[s]], axis=1)
  y = (3 * np.sin(2 * np.pi * t / 4.0) + np.sin(2 * np.pi * t / 52.1775) + 2 * lat[s] * lon[s] +
       0.5 * rng.standard_normal(t.size))
  return X, y, [T - 1.0, 1.0, 1.0]


def _net(scales, width=512, depth=2, obs='NORMAL'):
  return NetSpec(width=width, depth=depth, input_scales=scales, fourier_degrees=[5, 5, 5],
                 interactions=[], seasonality_periods=[4.0, 52.1775], num_seasonal_harmonics=[2, 10],
                 observation_model=obs)


def _engine(net, X, y, **kw):
  from bayesnf_amd.engine import Engine
  return Engine(net, X=X, y=y, **kw)


def _fit(net, X, y, steps, **kw):
  eng = _engine(net, X, y, **kw)
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  losses = eng.train(0, steps)
  torch.cuda.synchronize()
  out = eng.get_params().copy(), losses.cpu().numpy()
  eng.close()
  return out


def test_c2_shard_invariance_and_loss_decrease_bf16():
  """C2 (N=10,232, F=57, W=512, depth 2, bf16): 8 members on one handle == 2 x 4 members on two
  handles with member offsets (what two ranks would run), and the loss falls for every member."""
  X, y, scales = _grid()
  assert X.shape[0] == 10232
  net = _net(scales)
  assert net.F == 57
  kw = dict(seed=7, learning_rate=0.005, compute_dtype='bf16')
  th_all, loss_all = _fit(net, X, y, 12, members=8, **kw)
  th_a, loss_a = _fit(net, X, y, 12, members=4, member_offset=0, **kw)
  th_b, loss_b = _fit(net, X, y, 12, members=4, member_offset=4, **kw)
  # same random streams, same arithmetic; only the order of f32 atomics differs
  np.testing.assert_allclose(np.concatenate([loss_a, loss_b]), loss_all, rtol=2e-4)
  err = np.abs(np.concatenate([th_a, th_b]) - th_all).max()
  assert err < 5e-3, err
  assert np.all(np.isfinite(loss_all)) and np.all(loss_all[:, -1] < loss_all[:, 0])
  assert np.all(np.diff(loss_all, axis=1)[:, :6] < 0)


def test_c2_bf16_tracks_fp32_at_full_size():
  X, y, scales = _grid()
  net = _net(scales)
  kw = dict(seed=3, learning_rate=0.005, members=4)
  th32, l32 = _fit(net, X, y, 10, compute_dtype='fp32', **kw)
  th16, l16 = _fit(net, X, y, 10, compute_dtype='bf16', **kw)
  np.testing.assert_allclose(l16[:, 0], l32[:, 0], rtol=3e-3)     # same init, one forward
  np.testing.assert_allclose(l16[:, -1], l32[:, -1], rtol=2e-2)
  # the update direction agrees: parameters moved the same way
  d32, d16 = th32 - th32.mean(0), th16 - th16.mean(0)
  assert np.abs(th16 - th32).max() < 0.05


def test_c2_bf16_predictive_rmse_within_2pct_of_fp32():
  """SURVEY 8(d) bf16 gate: from identical initial parameters, 200 full-batch Adam steps at the C2 size
  (N = 10,232, F = 57, W = 512, depth 2) in bf16 (row-panel kernel) and in fp32: final loss within
  1 %, predictive RMSE of the ensemble-mean prediction on the training rows within 2 %; member-level
  statistics with the bars their measured run-to-run spread supports (see below)."""
  from bayesnf_amd.engine import Engine
  X, y, scales = _grid()
  net = _net(scales)
  kw = dict(seed=13, learning_rate=0.005, members=8)
  th32, l32 = _fit(net, X, y, 200, compute_dtype='fp32', **kw)
  th16, l16 = _fit(net, X, y, 200, compute_dtype='bf16', **kw)
  np.testing.assert_allclose(l16[:, 0], l32[:, 0], rtol=3e-3)        # same init
  np.testing.assert_allclose(l16[:, -1], l32[:, -1], rtol=1e-2)
  assert np.all(l32[:, -1] < 0.9 * l32[:, 0])                         # it did train (the loss carries the prior term)
  fwd = Engine(net, members=8, forward_only=True, row_capacity=4096, compute_dtype='fp32')
  Xd = torch.tensor(X, dtype=torch.float32, device=fwd.device)
  rmse = {}
  for name, th in (('fp32', th32), ('bf16', th16)):
    loc, _ = fwd.forward(torch.tensor(th, dtype=torch.float32, device=fwd.device), Xd)
    torch.cuda.synchronize()
    pred = loc.cpu().numpy()
    rmse[name] = (np.sqrt(np.mean((pred - y[None, :]) ** 2, axis=1)),
                  np.sqrt(np.mean((pred.mean(axis=0) - y) ** 2)))
  fwd.close()
  # What is robust and what is not was measured over 60 bf16 and 15 fp32 fits of this very problem
  # (scripts/rmse_gate_probe.py): two fp32 fits already differ by up to 2 % per member (f32 atomics
  # order + 200 Adam steps); bf16: final loss <= 0.1 %, RMSE of the ensemble-mean prediction <= 0.7 %,
  # mean over members <= 2.3 %, while a single member (the same one or two of the eight) lands 5-17 %
  # high in one run out of twelve.  Gates: loss 1 %, ensemble RMSE 2 % (the SURVEY 8(d) gate), median
  # member 3 %, mean over members 5 %, at most one member beyond 8 % and none beyond 20 %.
  # Round 3 (scripts/bf16_outlier_probe.py, profiles/r03_bf16_outliers.md; 192 bf16 and 160 fp32 fits): the
  # spread is the problem's, not the kernels' -- fp32 fits from identical initial parameters already differ by
  # 1-4 % in the same few members after 200 steps, the objective of an "outlier" member tracks the reference's
  # to 5e-4 at every checkpoint, and the first leaves to wander are the input-scale leaves; bf16 widens the
  # spread 2-3x: 20 of 192 runs with a member beyond 5 %, 2 beyond 8 %, largest 13.7 %.
  np.testing.assert_allclose(rmse['bf16'][1], rmse['fp32'][1], rtol=2e-2)
  np.testing.assert_allclose(np.median(rmse['bf16'][0]), np.median(rmse['fp32'][0]), rtol=3e-2)
  np.testing.assert_allclose(rmse['bf16'][0].mean(), rmse['fp32'][0].mean(), rtol=5e-2)
  dev = np.abs(rmse['bf16'][0] / rmse['fp32'][0] - 1.0)
  assert np.sum(dev > 0.08) <= 1 and dev.max() < 0.20, dev


@pytest.mark.parametrize('width,depth', [(512, 4), (768, 2), (1024, 2)])
def test_deep_and_wide_minibatch_mle(width, depth):
  """C4-shaped (minibatch MLE, depth 4 / widths the full-width last-layer kernel does not cover)."""
  X, y, scales = _grid(T=1200, S=20, seed=5)        # 23k rows
  net = _net(scales, width=width, depth=depth)
  th, losses = _fit(net, X, y, 3, members=2, batch=4096, prior_weight=0.0, seed=11,
                    learning_rate=0.005, compute_dtype='bf16')
  assert losses.shape == (2, 3) and np.all(np.isfinite(losses)) and np.all(np.isfinite(th))
  assert np.all(losses[:, -1] < losses[:, 0])


def test_c3_shaped_vi_depth4():
  """C3-shaped: mean-field VI, depth 4, W=512, B=3500, S=5."""
  X, y, scales = _grid(T=1000, S=20, seed=9)
  net = _net(scales, width=512, depth=4)
  eng = _engine(net, X, y, mode='vi', members=2, batch=3500, vi_samples=5, kl_weight=0.2, seed=2,
                learning_rate=0.01, compute_dtype='bf16')
  eng.init_params(0.0)
  losses = eng.train(0, 12)
  torch.cuda.synchronize()
  l = losses.cpu().numpy()
  mu_rho = eng.get_params()
  eng.close()
  assert l.shape == (2, 12) and np.all(np.isfinite(l)) and np.all(np.isfinite(mu_rho))
  # the ELBO estimate is stochastic (S = 5 draws per step): compare averages, not single epochs
  assert l[:, -3:].mean() < l[:, :3].mean()


@pytest.mark.parametrize('obs', ['NB', 'ZINB'])
def test_c2_count_models_full_size(obs):
  X, y, scales = _grid()
  counts = np.random.default_rng(1).poisson(np.exp(0.5 * y)).astype(np.float64)
  net = _net(scales, obs=obs)
  th, losses = _fit(net, X, counts, 8, members=2, seed=1, learning_rate=0.005, compute_dtype='bf16')
  assert np.all(np.isfinite(losses)) and np.all(losses[:, -1] < losses[:, 0])


def test_forward_only_bf16_matches_fp32_at_width_512():
  """Predict path (forward-only handle, 256 x 256 forward tiles + fused output row dot in bf16) against
  the fp32 path on the same parameters, in member and row chunks."""
  from bayesnf_amd.engine import Engine
  X, y, scales = _grid(T=300, S=20, seed=4)
  net = _net(scales, width=512, depth=2)
  rng = np.random.default_rng(0)
  M = 6
  theta = (0.3 * rng.standard_normal((M, net.P))).astype(np.float32)
  outs = {}
  for dt in ('fp32', 'bf16'):
    eng = Engine(net, members=4, forward_only=True, row_capacity=2048, compute_dtype=dt)
    loc, aux = eng.forward(torch.tensor(theta, device=eng.device), torch.tensor(X, dtype=torch.float32,
                                                                                device=eng.device))
    torch.cuda.synchronize()
    outs[dt] = (loc.cpu().numpy(), aux.cpu().numpy())
    eng.close()
  assert outs['bf16'][0].shape == (M, X.shape[0])
  np.testing.assert_allclose(outs['bf16'][1], outs['fp32'][1], rtol=1e-6)
  scale = np.abs(outs['fp32'][0]).max()
  assert np.abs(outs['bf16'][0] - outs['fp32'][0]).max() < 2e-2 * scale


@pytest.mark.parametrize('dtype', ['bf16', 'fp8'])
def test_repeated_step_is_reproducible_at_c2(dtype):
  """Race screen for the ring-buffered K loops (counted vmcnt + raw barriers, LDS-DMA -- in the fp8 W x W stream as
  inline assembly beside compiler-scheduled transpose reads): the same forward + backward at the benchmark size must
  reproduce itself up to the order of f32 atomics."""
  X, y, scales = _grid()
  net = _net(scales)
  eng = _engine(net, X, y, members=8, seed=1, compute_dtype=dtype)
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  loss0, g0 = eng.debug_loss_and_grad()
  scale = np.abs(g0).max(axis=1, keepdims=True)
  for _ in range(12):
    loss, g = eng.debug_loss_and_grad()
    assert np.abs(loss - loss0).max() <= 1e-5 * np.abs(loss0).max()
    assert (np.abs(g - g0) / scale).max() < 1e-4
  eng.close()


@pytest.mark.parametrize('layout', ['C2', 'C5'])
def test_fp8_tracks_bf16_at_full_size(layout):
  """compute_dtype 'fp8' at BASELINE.json's C2 and C5 (per-GPU share of the rows, a few members) shapes: the fp8 operand
  copies only feed the Dense-kernel gradients, so 12 Adam steps from the same initial parameters stay on the bf16
  trajectory -- every loss within 2e-3 of the bf16 run's, the last one lower than the first.  (C5: W = 256, 128 padded
  features, 71,000 rows = the split-K ring kernel + the generic fp8 kernel; C2: the ring kernel on the K = 64 MFMA +
  the skinny stream.)"""
  if layout == 'C2':
    X, y, scales = _grid()
    net = _net(scales)
  else:
    rng = np.random.default_rng(5)
    n = 71000
    t = rng.integers(0, 8760, n).astype(np.float64)
    lat, lon = rng.standard_normal(n), rng.standard_normal(n)
    X = np.stack([t, lat, lon], axis=1)
    y = (np.sin(2 * np.pi * t / 24.0) + 0.5 * np.sin(2 * np.pi * t / 168.0) + lat * lon + 0.3 * rng.standard_normal(n))
    net = NetSpec(width=256, depth=2, input_scales=[8759.0, 1.0, 1.0], fourier_degrees=[5, 5, 5], interactions=[],
                  seasonality_periods=[24.0, 168.0, 8766.0], num_seasonal_harmonics=[6, 8, 9], observation_model='NORMAL')
  kw = dict(members=4, seed=3, learning_rate=0.005)
  th16, l16 = _fit(net, X, y, 12, compute_dtype='bf16', **kw)
  th8, l8 = _fit(net, X, y, 12, compute_dtype='fp8', **kw)
  assert np.all(np.isfinite(l8)) and np.all(l8[:, -1] < l8[:, 0])
  assert np.abs(l8 / l16 - 1).max() < 2e-3, np.abs(l8 / l16 - 1).max()
