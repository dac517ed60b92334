"""End-to-end through the public estimator API on the GPU, including the
reference's own fixture (KAT K1-K4 of SURVEY.md, now with the HIP engine)."""
import os

import numpy as np
import pandas as pd
import pytest

from bayesnf_amd import BayesianNeuralFieldMAP, BayesianNeuralFieldMLE, BayesianNeuralFieldVI
from oracle import bnf_oracle as O

pytestmark = pytest.mark.gpu

MODEL = dict(width=256, depth=2, seasonality_periods=np.asarray([4.0, 52.1775]),
             num_seasonal_harmonics=np.asarray([2.0, 10]), observation_model='NORMAL',
             feature_cols=['datetime', 'latitude', 'longitude'], target_col='chickenpox',
             timetype='index', freq='W', standardize=['latitude', 'longitude'])


def _train_frame(golden_dir):
  return pd.read_csv(os.path.join(golden_dir, 'chickenpox.8.train.csv'), index_col=0,
                     parse_dates=['datetime'])


@pytest.mark.parametrize('cls,gold_name,gold_hw', [
    (BayesianNeuralFieldMAP, 'bnf-map.chickenpox.8.mini.pred.csv', 37.9523),
    (BayesianNeuralFieldMLE, 'bnf-mle.chickenpox.8.mini.pred.csv', 37.9533)])
def test_chickenpox_mini_map_mle(golden_dir, cls, gold_name, gold_hw):
  df = _train_frame(golden_dir)
  gold = pd.read_csv(os.path.join(golden_dir, gold_name), index_col=0).iloc[:100]
  est = cls(**MODEL).fit(df, seed=0, ensemble_size=4, num_epochs=5, learning_rate=0.005)
  assert est.losses_.shape == (1, 4, 5) and np.all(np.diff(est.losses_, axis=-1) < 0)
  assert len(est.params_) == 19 and est.params_.var4.shape == (1, 4, 57, 256)
  assert est.params_[0].shape == (1, 4) and est.params_._fields[0] == 'var0'
  means, qs = est.predict(df, quantiles=(0.5, 0.025, 0.975))
  assert means.shape == (1, 4, 100) and len(qs) == 3 and qs[0].shape == (100,)
  hw = ((qs[2] - qs[1]) / 2).mean()
  np.testing.assert_allclose(hw, gold_hw, rtol=2e-4)
  yhat = means.mean(axis=(0, 1))
  assert 0.5 * gold.yhat.mean() < yhat.mean() < 1.6 * gold.yhat.mean()
  # the engine's own parameters, pushed through the oracle, give the same prediction
  from bayesnf_amd.spec import NetSpec
  X = est.data_handler.get_test(df)
  args = est._model_args(X.shape)
  model = O.Model(**{k: args[k] for k in ('width', 'depth', 'input_scales', 'fourier_degrees',
                                          'interactions', 'seasonality_periods',
                                          'num_seasonal_harmonics')})
  net = NetSpec(**args)
  theta = net.pack(list(est.params_))[0]
  mu_o, sd_o = O.predict_normal(model, theta, X.astype(np.float32).astype(np.float64))
  np.testing.assert_allclose(means[0], mu_o, rtol=2e-3, atol=2e-4)
  for i, q in enumerate((0.5, 0.025, 0.975)):
    np.testing.assert_allclose(O.mixture_cdf(mu_o, sd_o, qs[i]), q, atol=1e-4)
  m2, q2 = est.predict(df, quantiles=(0.9,), approximate_quantiles=True)
  np.testing.assert_allclose(q2[0], O.approximate_normal_quantile(mu_o, sd_o, 0.9), rtol=1e-3)
  lik = est.likelihood_model(df)
  assert lik.mean().shape == (1, 4, 100) and lik.log_prob(df['chickenpox'].values).shape == (1, 4)


@pytest.mark.parametrize('dtype', ['fp32', None])     # the exact f32 chain; the estimators' DEFAULT (None -> 'fp32_split')
@pytest.mark.parametrize('cls,gold_name', [
    (BayesianNeuralFieldMAP, 'bnf-map.chickenpox.8.mini.pred.csv'),
    (BayesianNeuralFieldMLE, 'bnf-mle.chickenpox.8.mini.pred.csv')])
def test_reference_golden_reproduced_elementwise_through_the_engine(golden_dir, cls, gold_name, dtype, monkeypatch):
  """N1: `fit(seed=PRNGKey(0))` starts from the reference's own initial parameters (threefry + TFP
  seed chain, bayesnf_amd/jaxseed.py), so the fp32 HIP engine reproduces the reference's golden
  predictions of the training rows ELEMENT-WISE (reference test tests/test_evaluate_mini.py:58-78:
  4 particles, 5 epochs, lr 0.005, full batch, quantiles .5/.025/.975).  fp32 bar: 1e-4 absolute
  on yhat (the oracle itself is 2e-6 / 5e-6 from the golden); quantile columns are valid roots of
  the mixture CDF within the reference's value tolerance and within 5e-3 of the golden iterate."""
  monkeypatch.delenv('BNF_DTYPE', raising=False)
  df = _train_frame(golden_dir)
  gold = pd.read_csv(os.path.join(golden_dir, gold_name), index_col=0).iloc[:100]
  est = cls(**MODEL, compute_dtype=dtype).fit(df, seed=np.array([0, 0], dtype=np.uint32), ensemble_size=4,
                                             num_epochs=5, learning_rate=0.005)
  from bayesnf_amd import engine as _engine_mod
  assert _engine_mod.default_dtype(dtype) == ('fp32' if dtype else 'fp32_split')
  means, qs = est.predict(df, quantiles=(0.5, 0.025, 0.975))
  yhat = means.mean(axis=(0, 1))
  assert np.abs(yhat - gold.yhat.values).max() < 1e-4, np.abs(yhat - gold.yhat.values).max()
  sd = np.exp(np.asarray(est.params_[0], dtype=np.float64)) + 0.01           # (1, 4) noise scales
  for col, q, got in [('yhat_p50', 0.5, qs[0]), ('yhat_lower', 0.025, qs[1]), ('yhat_upper', 0.975, qs[2])]:
    g = gold[col].values
    assert np.abs(got - g).max() < 5e-3, (col, np.abs(got - g).max())
    assert np.abs(O.mixture_cdf(means, sd, g) - q).max() < 1.5e-5, col
    assert np.abs(O.mixture_cdf(means, sd, got) - q).max() < 2e-5, col
  # the device generator is a different stream: same statistics, not the same numbers
  est_p = cls(**MODEL, compute_dtype='fp32', init_rng='philox').fit(df, seed=0, ensemble_size=4, num_epochs=5,
                                                                    learning_rate=0.005)
  m_p, _ = est_p.predict(df, quantiles=(0.5,))
  assert np.abs(m_p.mean(axis=(0, 1)) - gold.yhat.values).max() > 1e-3
  # bf16 engine from the same initial parameters: the statistical class
  est_b = cls(**MODEL, compute_dtype='bf16').fit(df, seed=0, ensemble_size=4, num_epochs=5, learning_rate=0.005)
  m_b, _ = est_b.predict(df, quantiles=(0.5,))
  assert np.abs(m_b.mean(axis=(0, 1)) - gold.yhat.values).max() < 5e-2


def test_chickenpox_mini_vi(golden_dir):
  df = _train_frame(golden_dir)
  gold = pd.read_csv(os.path.join(golden_dir, 'bnf-vi.chickenpox.8.mini.pred.csv'),
                     index_col=0).iloc[:100]
  hw_gold = ((gold.yhat_upper - gold.yhat_lower) / 2).mean()
  hws = []
  for seed in range(4):
    est = BayesianNeuralFieldVI(**MODEL).fit(
        df, seed=seed, ensemble_size=1, num_epochs=2, learning_rate=0.01, kl_weight=0.1,
        sample_size_divergence=5, sample_size_posterior=30)
    assert est.losses_.shape == (1, 1, 2)
    assert est.params_.var4.shape == (1, 30, 1, 57, 256)
    means, qs = est.predict(df, quantiles=(0.5, 0.025, 0.975))
    assert means.shape == (1, 30, 1, 100)
    hws.append(((qs[2] - qs[1]) / 2).mean())
  assert min(hws) - 0.3 < hw_gold < max(hws) + 0.3, (hws, hw_gold)


def test_minibatch_and_splits_api(golden_dir):
  df = _train_frame(golden_dir)
  est = BayesianNeuralFieldMAP(**{**MODEL, 'width': 64}).fit(
      df, seed=[0, 42], ensemble_size=4, num_epochs=3, batch_size=32, num_splits=2)
  assert est.losses_.shape == (1, 4, 3) and est.params_.var6.shape == (1, 4, 64, 64)
  # the two splits use different seeds -> different members
  assert not np.allclose(est.params_.var4[0, 0], est.params_.var4[0, 2])
  with pytest.raises(ValueError):
    BayesianNeuralFieldMAP(**MODEL).fit(df, seed=0, ensemble_size=0, num_epochs=1)


def test_experiment_driver_writes_reference_file_layout(golden_dir, tmp_path):
  """run_experiment on the reference's own fixture (tests/test_evaluate_mini.py:58-67 config):
  same three output files and columns; half-width matches the golden (KAT K1)."""
  from bayesnf_amd import evaluate as ev
  icfg = {'num_particles': 4, 'num_epochs': 5, 'learning_rate': 0.005}
  dcfg = ev.DATASET_CONFIG['chickenpox']
  losses, means, qs = ev.run_experiment('chickenpox', golden_dir, '8', str(tmp_path), 'map', dcfg,
                                        ev.MODEL_CONFIG['chickenpox']['map'], icfg, seed=0)
  stem = tmp_path / 'bnf-map.chickenpox.8'
  pred = pd.read_csv(str(stem) + '.pred.csv', index_col=0)
  gold = pd.read_csv(os.path.join(golden_dir, 'bnf-map.chickenpox.8.mini.pred.csv'), index_col=0)
  assert list(pred.columns) == list(gold.columns) and list(pred.index) == list(gold.index)
  loss = pd.read_csv(str(stem) + '.loss.csv')
  assert loss.shape == (5, 4) and os.path.exists(str(stem) + '.log.json')
  hw = ((pred.yhat_upper - pred.yhat_lower) / 2).iloc[:100].mean()
  np.testing.assert_allclose(hw, 37.9523, rtol=2e-4)
  assert losses.shape == (1, 4, 5) and means.shape == (1, 4, 308) and qs.shape == (3, 308)


@pytest.mark.parametrize('obs', ['NB', 'ZINB'])
def test_count_observation_models_end_to_end(golden_dir, obs):
  """chickenpox counts under the NB / ZINB likelihoods (models.py:166-191) through fit / predict /
  likelihood_model: shapes, integer quantiles in order, loss decreases."""
  df = _train_frame(golden_dir)
  kw = dict(MODEL, observation_model=obs, width=64)
  est = BayesianNeuralFieldMAP(**kw).fit(df, seed=3, ensemble_size=4, num_epochs=60, learning_rate=0.01)
  assert est.losses_.shape == (1, 4, 60) and np.all(est.losses_[..., -1] < est.losses_[..., 0])
  means, qs = est.predict(df, quantiles=(0.5, 0.025, 0.975))
  assert means.shape == (1, 4, 100) and np.all(means > 0)
  lo, mid, hi = qs[1], qs[0], qs[2]
  assert all(np.all(q == np.round(q)) and np.all(q >= 0) for q in qs)
  assert np.all(lo <= mid) and np.all(mid <= hi) and np.any(hi > lo)
  lik = est.likelihood_model(df)
  np.testing.assert_allclose(lik.mean(), means, rtol=1e-4)
  assert lik.log_prob(df['chickenpox'].to_numpy()).shape == (1, 4)
  vi = BayesianNeuralFieldVI(**kw).fit(df, seed=1, ensemble_size=2, num_epochs=5, sample_size_posterior=3)
  m2, q2 = vi.predict(df, quantiles=(0.5,))
  assert m2.shape == (1, 3, 2, 100) and np.all(np.isfinite(m2)) and np.all(q2[0] >= 0)


def test_posterior_gather_through_the_c_abi_single_rank():
  """bnf_allgather (include/bnf.h): RCCL all-gather behind the C ABI, here with a one-rank
  communicator on the one GPU a test box has (the N-rank path is the same call; RCCL's ring over
  xGMI needs >= 2 devices, which only the driver's scaling run has)."""
  import torch
  from bayesnf_amd import _native
  dev = torch.device('cuda:0')
  send = torch.arange(3 * 1000, dtype=torch.float32, device=dev).reshape(3, 1000)
  recv = torch.zeros((1, 3, 1000), dtype=torch.float32, device=dev)
  _native.allgather(send, recv, world=1, rank=0)
  torch.cuda.synchronize(dev)
  assert torch.equal(recv[0], send)
  # a second call reuses the cached communicator
  send2 = send * 2
  _native.allgather(send2, recv, world=1, rank=0)
  torch.cuda.synchronize(dev)
  assert torch.equal(recv[0], send2)


def test_local_communicator_set_and_grouped_gather_single_device():
  """bnf_comm_create_local / bnf_allgather_group (include/bnf.h): what ONE process driving several GPUs gathers its shards
  with (`distributed.gather_shards`) -- here the set of one device a test box has: ncclCommInitAll, one grouped
  ncclAllGather, the set cached per device list; a device named twice is refused (and the callers fall back to peer
  copies).  Also: the library binds the RCCL the process has already mapped (torch's), not a second copy."""
  import torch
  from bayesnf_amd import _native, distributed
  dev = torch.device('cuda:0')
  send = torch.arange(2 * 500, dtype=torch.float32, device=dev).reshape(2, 500)
  recv = torch.zeros((1, 2, 500), dtype=torch.float32, device=dev)
  _native.allgather_local([send], [recv])
  torch.cuda.synchronize(dev)
  assert torch.equal(recv[0], send)
  _native.allgather_local([send * 3], [recv])           # cached communicator set
  torch.cuda.synchronize(dev)
  assert torch.equal(recv[0], send * 3)
  with pytest.raises(RuntimeError, match='listed twice'):
    _native.allgather_local([send, send], [recv, recv])
  parts = [send, send + 1]                               # the same device twice: gather_shards falls back to peer copies
  out = distributed.gather_shards(parts)
  assert out.shape == (2, 2, 500) and torch.equal(out[1], send + 1) and distributed.last_gather()['impl'] == 'peer-copies'
  mapped = [l.split()[-1] for l in open('/proc/self/maps') if 'librccl' in l]
  assert len(set(mapped)) == 1, set(mapped)              # one RCCL in the process


@pytest.mark.parametrize('obs', ['NORMAL', 'NB', 'ZINB'])
def test_likelihood_model_against_the_oracle(golden_dir, obs):
  """N3: the object `likelihood_model(table)` returns (reference spatiotemporal.py:433-468 returns
  the TFP distribution) against the oracle evaluated on the fitted parameters: mean, stddev,
  log_prob, cdf per member, and the mixture quantile as a root of the oracle's mixture cdf."""
  from bayesnf_amd.spec import NetSpec
  df = _train_frame(golden_dir)
  kw = dict(MODEL, observation_model=obs, width=64, compute_dtype='fp32')
  est = BayesianNeuralFieldMAP(**kw).fit(df, seed=5, ensemble_size=3, num_epochs=40, learning_rate=0.01)
  lik = est.likelihood_model(df)
  X = est.data_handler.get_test(df).astype(np.float32).astype(np.float64)
  y = df['chickenpox'].to_numpy().astype(np.float64)
  args = est._model_args(X.shape)
  model = O.Model(observation_model=obs, **{k: args[k] for k in (
      'width', 'depth', 'input_scales', 'fourier_degrees', 'interactions', 'seasonality_periods',
      'num_seasonal_harmonics')})
  theta = NetSpec(observation_model=obs, **args).pack(list(est.params_))[0].astype(np.float64)
  out = O.forward(model, theta, X)
  if obs == 'NORMAL':
    sigma = O.noise_scale(model, theta)
    np.testing.assert_allclose(lik.mean()[0], out, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(lik.stddev()[0], np.broadcast_to(sigma[:, None], out.shape), rtol=1e-5)
    np.testing.assert_allclose(lik.log_prob(y)[0], O.normal_loglik(out, y, sigma), rtol=2e-5)
    xs = np.linspace(y.min(), y.max(), 100)
    np.testing.assert_allclose(lik.cdf(xs)[0], O._ndtr((xs[None, :] - out) / sigma[:, None]), atol=2e-5)
    np.testing.assert_allclose(lik.mixture_cdf(xs), O.mixture_cdf(out, sigma, xs), atol=2e-5)
    for q in (0.1, 0.5, 0.9):
      np.testing.assert_allclose(O.mixture_cdf(out, sigma, lik.quantile(q)), q, atol=3e-5)
    assert lik.quantile([0.25, 0.75]).shape == (2, 100)
  else:
    fc = O.count_forecast(model, theta, out)
    np.testing.assert_allclose(lik.mean()[0], fc['mean'], rtol=5e-4)
    np.testing.assert_allclose(lik.stddev()[0], fc['stddev'], rtol=5e-4)
    tc, logits = O.nb_logits_total_count(model, theta, out)
    if obs == 'NB':
      lp = O.nb_log_prob(y[None, :], tc, logits).sum(axis=-1)
    else:
      lp = O.zinb_log_prob(y[None, :], tc, logits, fc['pi']).sum(axis=-1)
    np.testing.assert_allclose(lik.log_prob(y)[0], lp, rtol=2e-4)
    xs = np.arange(0, 100, dtype=np.float64)
    np.testing.assert_allclose(lik.cdf(xs)[0], O.count_cdf(fc, xs[None, :]), atol=2e-4)
    for q in (0.2, 0.5, 0.9):
      got = lik.quantile(q)
      np.testing.assert_array_equal(got, O.count_quantile_via_root(fc, q))
      # an integer quantile: cdf(k) >= q > cdf(k - 1) up to the root tolerance
      assert np.all(lik.mixture_cdf(got) >= q - 2e-5)


def test_reference_vi_golden_reproduced_through_the_engine(golden_dir):
  """N1, VI on the GPU: the fp32 engine, started from the reference's initial surrogate means
  (bayesnf_amd/jaxseed.vi_initial_means = what BayesianNeuralFieldVI.fit uses) and fed the
  reparameterisation noise of the reference's two optimisation steps and of its 30 posterior draws
  (oracle/jax_rng.py, through the verification hook bnf_debug_vi_noise), reproduces column `yhat` of
  bnf-vi.chickenpox.8.mini.pred.csv element-wise.  (In production the engine draws that noise from
  its own counter-based generator: same law, other numbers.)"""
  import torch
  from bayesnf_amd import jaxseed
  from bayesnf_amd.engine import Engine
  from oracle import jax_rng as R
  from tests.test_oracle_kat import _setup
  from bayesnf_amd import spatiotemporal as st
  model, X, y = _setup(golden_dir, st.BayesianNeuralFieldVI)
  gold = pd.read_csv(os.path.join(golden_dir, 'bnf-vi.chickenpox.8.mini.pred.csv'), index_col=0).iloc[:100]
  df = _train_frame(golden_dir)
  est = BayesianNeuralFieldVI(**MODEL, compute_dtype='fp32')
  Xe = est.data_handler.get_train(df)
  from bayesnf_amd import inference as bnf_inference
  net = bnf_inference._net_from_args(est._model_args(Xe.shape), 'NORMAL')
  seed = R.prng_key(0)
  E, S, steps, draws = 1, 5, 2, 30
  mu0 = jaxseed.vi_initial_means(net, seed, 1, E)[0]
  np.testing.assert_array_equal(mu0, R.reference_vi_init_means(model, seed, E).astype(np.float32))
  eng = Engine(net, mode='vi', X=X, y=y, members=E, vi_samples=S, kl_weight=0.1, learning_rate=0.01,
               seed=0, compute_dtype='fp32')
  eng.init_params(0.0)
  p0 = eng.get_params()
  p0[0] = mu0
  eng.set_params(p0)
  noise = R.reference_vi_step_noise(model, seed, steps, S, E)
  for s in range(steps):
    eng.debug_vi_noise(torch.tensor(noise[s]))
    eng.train(s, 1)
    torch.cuda.synchronize()
  eng.debug_vi_noise(None)
  mu, rho = eng.get_params().astype(np.float64)
  eng.close()
  eps = R.reference_vi_posterior_noise(model, seed, draws, E)
  theta = (mu[None] + O.vi_sigma(rho)[None] * eps).reshape(draws * E, model.P)
  fwd = Engine(net, members=draws, forward_only=True, row_capacity=128, compute_dtype='fp32')
  loc, aux = fwd.forward(torch.tensor(theta, dtype=torch.float32, device=fwd.device),
                         torch.tensor(X, dtype=torch.float32, device=fwd.device))
  torch.cuda.synchronize()
  yhat = loc.cpu().numpy().mean(axis=0)
  fwd.close()
  assert np.abs(yhat - gold.yhat.values).max() < 1e-4, np.abs(yhat - gold.yhat.values).max()


def test_vi_fit_reproduces_the_reference_golden(golden_dir):
  """The product path, no hooks: BayesianNeuralFieldVI.fit(seed=PRNGKey(0)) with the reference's mini
  configuration (tests/test_evaluate_mini.py:81-91) draws the reference's initial surrogate means, its
  optimisation noise and its posterior draws (keys on the host -- jaxseed --, jax.random.normal restated on
  the device: threefry2x32 + erfinv) and reproduces bnf-vi.chickenpox.8.mini.pred.csv element-wise."""
  import torch
  from oracle import jax_rng as R
  from tests.test_oracle_kat import _setup
  from bayesnf_amd import spatiotemporal as st
  df = _train_frame(golden_dir)
  gold = pd.read_csv(os.path.join(golden_dir, 'bnf-vi.chickenpox.8.mini.pred.csv'), index_col=0).iloc[:100]
  est = BayesianNeuralFieldVI(**MODEL, compute_dtype='fp32').fit(
      df, seed=np.array([0, 0], dtype=np.uint32), ensemble_size=1, num_epochs=2, learning_rate=0.01,
      kl_weight=0.1, sample_size_divergence=5, sample_size_posterior=30)
  means, qs = est.predict(df, quantiles=(0.5, 0.025, 0.975))
  yhat = np.asarray(means).mean(axis=(0, 1, 2))
  assert np.abs(yhat - gold.yhat.values).max() < 1e-4, np.abs(yhat - gold.yhat.values).max()
  for col, got in [('yhat_p50', qs[0]), ('yhat_lower', qs[1]), ('yhat_upper', qs[2])]:
    assert np.abs(got - gold[col].values).max() < 5e-3, col
  # the device generator against the oracle's restatement of jax.random.normal, value by value
  from bayesnf_amd import jaxseed, inference as bnf_inference
  from bayesnf_amd.engine import Engine
  model, X, y = _setup(golden_dir, st.BayesianNeuralFieldVI)
  net = bnf_inference._net_from_args(est._model_args(est.data_handler.get_train(df).shape), 'NORMAL')
  seed = R.prng_key(0)
  for E in (1, 3):
    eng = Engine(net, mode='vi', X=X, y=y, members=E, vi_samples=5, kl_weight=0.1, learning_rate=0.01, seed=0,
                 compute_dtype='fp32')
    eng.init_params(0.0)
    eng.set_vi_noise_keys(jaxseed.vi_noise_keys(net, seed, 1, 0, 2, 5), jaxseed.vi_draw_keys(net, seed, 1, 0, 7),
                          jaxseed.leaf_offsets(net))
    ref = R.reference_vi_step_noise(model, seed, 2, 5, E)
    for s in range(2):
      # (float32 erfinv polynomial on the device vs scipy's erfinv in the oracle: relative 6e-6 in the tails)
      np.testing.assert_allclose(eng.debug_vi_eps(s), ref[s], rtol=2e-5, atol=3e-6)
    p = eng.get_params().astype(np.float64)
    draws = eng.vi_posterior_draws(7).cpu().numpy()
    eps = R.reference_vi_posterior_noise(model, seed, 7, E)
    np.testing.assert_allclose(draws, p[0][None] + O.vi_sigma(p[1])[None] * eps, rtol=2e-5, atol=2e-5)
    eng.close()


@pytest.mark.parametrize('cls,pw', [(BayesianNeuralFieldMAP, 1.0), (BayesianNeuralFieldMLE, 0.0)])
def test_minibatch_fit_follows_the_reference_shuffle_stream(golden_dir, cls, pw):
  """`fit(seed, batch_size=32)` through the product path (jaxseed key chain -> bnf_row_tables -> fp32
  engine) equals the oracle trained on the reference's own per-member per-epoch
  `jax.random.permutation` stream (oracle/jax_rng.py restates /root/reference/src/bayesnf/
  inference.py:35-39,571-575,593-597) from the reference's own initial particles: minibatch fits are
  seed-for-seed like full-batch ones (which the reference's goldens pin).  100 rows, batch 32: three
  steps per epoch, the ragged tail of 4 rows dropped."""
  from oracle import jax_rng as R
  from tests.test_oracle_kat import _setup
  from bayesnf_amd import spatiotemporal as st
  df = _train_frame(golden_dir)
  E, epochs, B = 4, 3, 32
  est = cls(**MODEL).fit(df, seed=0, ensemble_size=E, num_epochs=epochs, learning_rate=0.005, batch_size=B)
  model, X, y = _setup(golden_dir, st.BayesianNeuralFieldMAP)
  key = R.prng_key(0)
  theta0 = O.map_init(model, y, R.reference_map_init_matrices(model, key, E), dtype=np.float32).astype(np.float64)
  perms = R.reference_map_permutations(key, E, epochs, len(y))             # (E, epochs, N)
  theta_o, losses_o = O.train_map(model, theta0, X, y, lr=0.005,
                                  num_epochs=epochs, batch_size=B, prior_weight=pw,
                                  row_index_fn=lambda ep: perms[:, ep, :(len(y) // B) * B])
  np.testing.assert_allclose(est.losses_[0], losses_o, rtol=1e-4)
  from bayesnf_amd.spec import NetSpec
  args = est._model_args(est.data_handler.get_test(df).shape)
  theta_d = NetSpec(**args).pack(list(est.params_))[0]
  assert util_rel_err(theta_d, theta_o) < 1e-3
  # and it is NOT the engine's own shuffle: the Feistel stream gives other numbers from the same start
  os.environ['BNF_INIT_RNG'] = 'philox'
  try:
    other = cls(**MODEL).fit(df, seed=0, ensemble_size=E, num_epochs=epochs, learning_rate=0.005, batch_size=B)
  finally:
    del os.environ['BNF_INIT_RNG']
  assert np.abs(other.losses_[0] - losses_o).max() > 1e-3 * np.abs(losses_o).max()


def util_rel_err(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


@pytest.mark.parametrize('mode', ['map', 'vi'])
def test_initial_parameters_drawn_on_the_device_equal_the_host_chain(mode):
  """bnf_init_params_keys: the reference's initial Dense kernels (TruncatedNormal(0, 1, -2, 2) through
  jax.random.truncated_normal: threefry bits -> uniform -> sqrt2 erfinv -> clip) drawn on the device from the
  members' per-leaf keys, against the host restatement that is pinned to the oracle and the goldens
  (jaxseed.map_initial_params / vi_initial_means).  The device evaluates the f32 erfinv polynomial XLA uses (what
  the reference itself runs), the host rounds scipy's f64 erfinv: the polynomial's own error, <= 1e-6 absolute on
  values in (-2, 2) (measured 9.5e-7; 87 % of the elements within one ulp)."""
  from bayesnf_amd import jaxseed as J
  from bayesnf_amd.engine import Engine
  from tests import util
  net, model, X, y = util.make_problem(n_rows=120, width=192, depth=3)
  E = 4
  if mode == 'map':
    keys = J.member_keys(7, 1, E, None)[0]
    want = J.map_initial_params(net, keys, 0.37)
    eng = Engine(net, X=X, y=y, members=E, seed=0)
    eng.init_params_keys(J.map_leaf_keys(net, keys), 0.37)
    got = eng.get_params()
  else:
    want = J.vi_initial_means(net, 7, 1, E)[0]
    eng = Engine(net, X=X, y=y, members=E, seed=0, mode='vi', vi_samples=2, kl_weight=0.1)
    eng.init_params_keys(J.vi_mean_leaf_keys(net, 7, 1, E)[0], 0.0)
    got, rho = eng.get_params()
    np.testing.assert_allclose(rho, -1.0502256128148466, rtol=1e-6)      # softplus^-1(0.3)
  eng.close()
  assert got.shape == want.shape
  np.testing.assert_allclose(got, want, rtol=0, atol=1.2e-6)
  assert np.abs(got).max() < 2.0
  nz = want != 0
  assert nz.mean() > 0.9 and np.array_equal(got == 0, want == 0)          # Dense kernels drawn, everything else 0
  lns = net.by_name['log_noise_scale'].offset
  if mode == 'map':
    assert np.all(got[:, lns] == np.float32(0.37))
