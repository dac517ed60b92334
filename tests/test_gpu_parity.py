"""Parity of the HIP engine (through the C ABI) against the CPU oracle.

The f32-class engines -- 'fp32' (BNF_DTYPE_F32: exact f32 MFMA contractions, the estimators' default) and 'fp32_split'
(BNF_DTYPE_F32S: f32 storage and epilogues, contractions as three bf16 MFMAs on operands split in registers) -- are both
held to SURVEY 8d's fp32 gates verbatim (tests/util.py FP32_GATE: forward 1e-5, loss 1e-5, every gradient leaf 1e-4 of the
leaf's max, parameters after 100 full-batch Adam steps 1e-3) against the float64 oracle that uses the reference's float32
trig arguments.  Transcendentals are the hardware's v_exp_f32 / v_rcp_f32 (1 ulp), reductions f32 atomics; the Fourier
features amplify a 1-ulp difference in the scaled input by up to 2 pi 2^4 -> feature errors up to ~2e-5 absolute (bar 5e-5
abs); intermediate activations 2e-4 rel; quantile CDF residual 2e-5.  bf16 engine: statistical.
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


# --------------------------------------------------------------------------- GEMM core
# The contraction cores on their own, max |C - ref| / max |ref|: 'fp32' (BNF_DTYPE_F32: v_mfma_f32_32x32x2_f32, exact
# products) and 'bf16' (against the product of the bf16-rounded operands) to f32 summation order; 'fp32_split'
# (BNF_DTYPE_F32S: f32 operands split in registers into two bf16 pieces, hi*hi + hi*lo + lo*hi on bf16 MFMAs: 16 operand
# bits) measured 3.6e-6 .. 5.9e-6 (gpurun_out/r05l) -- bar 1e-5.
_CORE_BAR = {'fp32': 2e-6, 'fp32_split': 1e-5, 'bf16': 2e-6}
G = util.FP32_GATE


@pytest.mark.parametrize('dtype', ['fp32', 'fp32_split', 'bf16'])
@pytest.mark.parametrize('shape', [(128, 128, 64), (300, 200, 128), (57, 512, 320), (1000, 64, 64)])
def test_contraction_core(dtype, shape):
  M, N, K = shape
  net, model, X, y = util.make_problem(n_rows=8, width=64, depth=1)
  eng = _engine(net, X, y, members=1, compute_dtype=dtype)
  rng = np.random.default_rng(M + N)
  A = rng.standard_normal((M, K)).astype(np.float32)
  Bt = rng.standard_normal((N, K)).astype(np.float32)   # asymmetric on purpose
  Cd = eng.debug_gemm_nt(A, Bt)
  if dtype == 'bf16':
    A = torch.tensor(A).bfloat16().float().numpy()
    Bt = torch.tensor(Bt).bfloat16().float().numpy()
  ref = A.astype(np.float64) @ Bt.astype(np.float64).T
  assert util.rel_err(Cd, ref) < _CORE_BAR[dtype], util.rel_err(Cd, ref)
  eng.close()


@pytest.mark.parametrize('dtype', ['fp32', 'fp32_split', 'bf16'])
@pytest.mark.parametrize('shape', [(64, 128, 128), (320, 64, 192), (1024, 512, 512), (128, 56, 200)])
def test_weight_gradient_core(dtype, shape):
  """C = A^T B on row-major operands (transpose reads in LDS)."""
  R, M, N = shape
  net, model, X, y = util.make_problem(n_rows=8, width=64, depth=1)
  eng = _engine(net, X, y, members=1, compute_dtype=dtype)
  rng = np.random.default_rng(R + M)
  A = rng.standard_normal((R, M)).astype(np.float32)
  B = rng.standard_normal((R, N)).astype(np.float32)
  Cd = eng.debug_gemm_tn(A, B)
  if dtype == 'bf16':
    A = torch.tensor(A).bfloat16().float().numpy()
    B = torch.tensor(B).bfloat16().float().numpy()
  ref = A.astype(np.float64).T @ B.astype(np.float64)
  assert util.rel_err(Cd, ref) < _CORE_BAR[dtype], util.rel_err(Cd, ref)
  eng.close()


@pytest.mark.parametrize('dtype', ['fp32', 'fp32_split', 'bf16'])
def test_large_tile_kernels_forced(dtype, monkeypatch):
  """256 x 256 tiles (16 waves) of the contraction cores, forced at sizes the oracle can check
  (production picks them only when they fill the chip): weight-gradient core + a train step."""
  monkeypatch.setenv('BNF_BIG_TILES', '2')
  net, model, X, y = util.make_problem(n_rows=64, width=64, depth=1)
  eng = _engine(net, X, y, members=1, compute_dtype=dtype)
  rng = np.random.default_rng(3)
  R, M, N = 640, 512, 256
  A = rng.standard_normal((R, M)).astype(np.float32)
  B = rng.standard_normal((R, N)).astype(np.float32)
  if dtype == 'bf16':
    A = torch.tensor(A).bfloat16().float().numpy()
    B = torch.tensor(B).bfloat16().float().numpy()
  Cd = eng.debug_gemm_tn(A, B)
  ref = A.astype(np.float64).T @ B.astype(np.float64)
  assert util.rel_err(Cd, ref) < _CORE_BAR[dtype], util.rel_err(Cd, ref)
  eng.close()
  # whole step at W = 256: forward 4 x 4 tiles (bf16), W x W weight gradient 4 x 4 tiles
  n_rows, E = 300, 2
  net, model, X, y = util.make_problem(n_rows=n_rows, width=256, depth=2)
  theta = util.random_theta(model, E)
  eng = _engine(net, X, y, members=E, compute_dtype=dtype, pipeline='layers')
  eng.set_params(theta)
  loss_d, g_d = eng.debug_loss_and_grad()
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  tol_l, tol_g = (G['loss'], G['grad']) if dtype.startswith('fp32') else (5e-3, 6e-2)
  np.testing.assert_allclose(loss_d, loss_o, rtol=tol_l)
  errs = util.per_leaf_rel_err(model, g_d, g_o)
  bad = {k: v for k, v in errs.items() if v > tol_g}
  assert not bad, bad
  eng.close()


# --------------------------------------------------------------------------- forward / grads
@pytest.mark.parametrize('depth,width,n_rows,pipeline', [
    (2, 64, 300, 'layers'), (1, 128, 130, 'layers'), (3, 192, 257, 'layers'), (2, 256, 200, 'layers'),
    (2, 64, 300, 'auto'), (1, 128, 130, 'auto'), (3, 256, 257, 'auto'), (2, 192, 140, 'auto')])
@pytest.mark.parametrize('dtype', util.FP32_DTYPES)
def test_forward_and_grad_fp32(depth, width, n_rows, pipeline, dtype):
  """The train-step pipelines: 'layers' = one kernel per layer, every activation materialised;
  'auto' (default) = the same with the last hidden layer, the output layer, the likelihood and
  its backward fused into one kernel where the width allows (64/128/256, 512 in bf16; 192
  falls back).  (The bf16 row-panel kernel has its own file, tests/test_gpu_panel.py.)"""
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  E = 3
  theta = util.random_theta(model, E)
  for pw in (1.0, 0.0):
    eng = _engine(net, X, y, members=E, prior_weight=pw, compute_dtype=dtype, pipeline=pipeline)
    eng.set_params(theta)
    loss_d, g_d = eng.debug_loss_and_grad()
    out_o, ch = O.forward(model, theta, X, keep=True)
    loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows, prior_weight=pw)
    H0 = eng.debug_activation(0)
    assert np.max(np.abs(H0 - ch['Hs'][0])) < 5e-5
    for l in range(depth):
      # the fused kernels keep (some) pre-activations on chip
      # ('auto' also recomputes the layer-0 pre-activation in the backward pass when depth >= 2)
      if pipeline == 'layers' or (pipeline == 'auto' and 0 < l and (l < depth - 1 or width == 192)):
        assert util.rel_err(eng.debug_activation(100 + l), ch['As'][l]) < 2e-4, l
      if l < depth - 1:   # the last hidden output is consumed in registers, never stored
        assert util.rel_err(eng.debug_activation(1 + l), ch['Hs'][l + 1]) < 2e-4, l
    assert util.rel_err(eng.debug_activation(200), out_o) < G['out']
    np.testing.assert_allclose(loss_d, loss_o, rtol=G['loss'])
    errs = util.per_leaf_rel_err(model, g_d, g_o)
    bad = {k: v for k, v in errs.items() if v > G['grad']}
    if bad:   # diagnostics: is it the prior term (g_d - g_o ~ -+ tanh(theta / 2)) and is it reproducible?
      d = g_d - g_o
      pr = np.tanh(0.5 * theta)
      loss_2, g_2 = eng.debug_loss_and_grad()
      bad = dict(bad, _pw=pw, _corr_with_prior=float(np.corrcoef(d.ravel(), pr.ravel())[0, 1]),
                 _second_eval_max_diff=float(np.abs(g_2 - g_d).max()),
                 _second_eval_bad=len([k for k, v in util.per_leaf_rel_err(model, g_2, g_o).items() if v > G['grad']]))
    assert not bad, bad
    eng.close()


@pytest.mark.parametrize('harmonics', [(40,), (90,)])
def test_many_seasonal_frequencies(harmonics):
  """Feature kernels with a wide seasonal block (up to 90 of the 96 supported frequencies)."""
  n_rows, E = 333, 2
  net, model, X, y = util.make_problem(n_rows=n_rows, width=64, depth=2, periods=(400.0,),
                                       harmonics=harmonics, T=1000)
  theta = util.random_theta(model, E)
  eng = _engine(net, X, y, members=E, compute_dtype='fp32')
  eng.set_params(theta)
  loss_d, g_d = eng.debug_loss_and_grad()
  out_o, ch = O.forward(model, theta, X, keep=True)
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  assert np.max(np.abs(eng.debug_activation(0) - ch['Hs'][0])) < 5e-5
  np.testing.assert_allclose(loss_d, loss_o, rtol=G['loss'])
  errs = util.per_leaf_rel_err(model, g_d, g_o)
  bad = {k: v for k, v in errs.items() if v > G['grad']}
  assert not bad, bad
  eng.close()


@pytest.mark.parametrize('width,pipeline', [(64, 'layers'), (64, 'auto'), (256, 'auto')])
@pytest.mark.parametrize('dtype', util.FP32_DTYPES)
def test_train_full_batch_fp32(width, pipeline, dtype):
  """parameters after 100 full-batch Adam steps within 1e-3 of the float64 oracle's (and the loss path 1e-5)."""
  n_rows, E, steps = 200, 4, 100
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=2)
  eng = _engine(net, X, y, members=E, seed=11, learning_rate=0.005, compute_dtype=dtype,
                pipeline=pipeline)
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  theta0 = eng.get_params().astype(np.float64)
  mm = model.matrix_mask()
  # init: kernels ~ TN(0,1,[-2,2]); everything else 0 except log_noise_scale
  assert np.all(np.abs(theta0[:, mm]) <= 2.0) and abs(theta0[:, mm].std() - 0.8796) < 0.03
  rest = theta0[:, ~mm].copy()
  rest[:, model.leaf['log_noise_scale'].offset - 0] = 0  # lns sits before any matrix
  assert np.all(rest == 0)
  np.testing.assert_allclose(theta0[:, model.leaf['log_noise_scale'].offset],
                             np.log(np.nanstd(y) / 2), rtol=1e-6)
  losses = eng.train(0, steps)
  torch.cuda.synchronize()
  theta_d = eng.get_params()
  theta_o, losses_o = O.train_map(model, theta0, X, y, lr=0.005, num_epochs=steps)
  np.testing.assert_allclose(losses.cpu().numpy(), losses_o, rtol=G['loss'])
  assert util.rel_err(theta_d, theta_o) < G['params100'], util.rel_err(theta_d, theta_o)
  eng.close()


@pytest.mark.parametrize('width', [64, 128])
def test_minibatch_shuffle_and_training_fp32(width):
  n_rows, B, E, epochs = 333, 100, 3, 2
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=2)
  eng = _engine(net, X, y, members=E, batch=B, seed=5, member_offset=7, compute_dtype='fp32')
  eng.init_params(0.3)
  theta0 = eng.get_params().astype(np.float64)
  steps = n_rows // B
  idx = {}
  for ep in range(epochs):
    rows = np.concatenate([eng.debug_row_index(ep, s) for s in range(steps)], axis=1)
    assert rows.shape == (E, steps * B)
    for e in range(E):  # a prefix of a permutation: no repeats, all in range
      assert len(set(rows[e].tolist())) == steps * B and rows[e].min() >= 0 and rows[e].max() < n_rows
    assert not np.array_equal(rows[0], rows[1])            # per-member shuffles differ
    idx[ep] = rows
  assert not np.array_equal(idx[0], idx[1])                # and change every epoch
  losses = eng.train(0, epochs)
  torch.cuda.synchronize()
  theta_o, losses_o = O.train_map(model, theta0, X, y, lr=0.005, num_epochs=epochs, batch_size=B,
                                  row_index_fn=lambda ep: idx[ep])
  np.testing.assert_allclose(losses.cpu().numpy(), losses_o, rtol=1e-4)
  assert util.rel_err(eng.get_params(), theta_o) < 1e-3
  # shard invariance: the same global members on a differently placed shard
  eng2 = _engine(net, X, y, members=1, batch=B, seed=5, member_offset=8, compute_dtype='fp32')
  eng2.init_params(0.3)
  np.testing.assert_array_equal(eng2.get_params()[0], theta0[1].astype(np.float32))
  np.testing.assert_array_equal(eng2.debug_row_index(1, 2)[0], eng.debug_row_index(1, 2)[1])
  eng.close(); eng2.close()


# --------------------------------------------------------------------------- VI
@pytest.mark.parametrize('width', [64, 128])
@pytest.mark.parametrize('dtype', util.FP32_DTYPES)
def test_vi_step_and_training_fp32(width, dtype):
  n_rows, E, S = 150, 2, 3
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=2)
  eng = _engine(net, X, y, mode='vi', members=E, vi_samples=S, kl_weight=0.2, seed=3,
                learning_rate=0.01, compute_dtype=dtype)
  eng.init_params(0.0)
  p0 = eng.get_params().astype(np.float64)
  mu0, rho0 = p0[0], p0[1]
  np.testing.assert_allclose(rho0, np.log(np.expm1(0.3)), rtol=1e-6)
  eps0 = eng.debug_vi_eps(0)
  assert eps0.shape == (E, S, model.P)
  assert abs(eps0.mean()) < 0.01 and abs(eps0.std() - 1) < 0.01
  assert abs(np.corrcoef(eps0[0, 0], eps0[0, 1])[0, 1]) < 0.02
  loss_d, g_d = eng.debug_loss_and_grad(0, 0)
  loss_o, gmu_o, grho_o = O.vi_loss_and_grad(model, mu0, rho0, eps0, X, y, n_rows, 0.2)
  np.testing.assert_allclose(loss_d, loss_o * 0.2, rtol=G['loss'])
  bad = {k: v for k, v in util.per_leaf_rel_err(model, g_d[0], gmu_o).items() if v > G['grad']}
  assert not bad, ('gmu', bad)
  bad = {k: v for k, v in util.per_leaf_rel_err(model, g_d[1], grho_o).items() if v > G['grad']}
  assert not bad, ('grho', bad)
  steps = 5
  eps = {s: eng.debug_vi_eps(s) for s in range(steps)}
  losses = eng.train(0, steps)
  torch.cuda.synchronize()
  mu_o, rho_o, losses_o = O.train_vi(model, mu0, rho0, X, y, lr=0.01, num_steps=steps,
                                     sample_size=S, kl_weight=0.2, eps_fn=lambda s: eps[s])
  np.testing.assert_allclose(losses.cpu().numpy(), losses_o, rtol=G['loss'])
  p = eng.get_params()
  assert util.rel_err(p[0], mu_o) < G['params100'] and util.rel_err(p[1], rho_o) < G['params100']
  draws = eng.vi_posterior_draws(7).cpu().numpy()
  assert draws.shape == (7, E, model.P)
  zs = (draws - p[0][None]) / O.vi_sigma(p[1].astype(np.float64))[None]
  assert abs(zs.mean()) < 0.01 and abs(zs.std() - 1) < 0.01
  eng.close()


@pytest.mark.parametrize('keep_z', ['1', '0'])
def test_vi_gradients_with_sigma_at_its_floor(keep_z, monkeypatch):
  """keep_z = 1 (BNF_VI_KEEP_Z=1; always the flow of the caller's / the reference's noise): k_vi_adam recovers the step's
  noise from the stored samples, eps = (z - mu) / sigma, instead of generating it a second time.  That quotient loses
  ulp(z) / sigma: with sigma at its floor (rho = -12: sigma = 1e-4 + softplus(-12) = 1.06e-4) and |mu| ~ 1 the recovered
  eps is off by ~6e-4 -- the worst case the parametrisation allows (sigma >= 1e-4).  The gradients wrt mu and rho must
  still meet bars next to the oracle's, which uses the exact noise.  keep_z = 0 (the default with the device generator since
  round 4): the noise is made again from the counter-based stream -- exact -- and meets the same bars."""
  monkeypatch.setenv('BNF_VI_KEEP_Z', keep_z)
  n_rows, E, S = 150, 2, 3
  net, model, X, y = util.make_problem(n_rows=n_rows, width=64, depth=2)
  eng = _engine(net, X, y, mode='vi', members=E, vi_samples=S, kl_weight=0.2, seed=3, learning_rate=0.01, compute_dtype='fp32')
  eng.init_params(0.0)
  p0 = eng.get_params().astype(np.float64)
  p0[1] = -12.0
  p0[0] = np.where(np.abs(p0[0]) < 0.5, np.sign(p0[0] + 1e-9) * 0.5 + p0[0], p0[0])     # |mu| >= 0.5 everywhere
  eng.set_params(p0)
  eps0 = eng.debug_vi_eps(0)
  loss_d, g_d = eng.debug_loss_and_grad(0, 0)
  loss_o, gmu_o, grho_o = O.vi_loss_and_grad(model, p0[0], p0[1], eps0, X, y, n_rows, 0.2)
  # (the entropy term -sum log sigma dominates this loss and goes through the hardware log2 / exp2: 2e-4, not the usual 5e-5)
  np.testing.assert_allclose(loss_d, loss_o * 0.2, rtol=2e-4)
  bad = {k: v for k, v in util.per_leaf_rel_err(model, g_d[0], gmu_o).items() if v > 5e-4}
  assert not bad, ('gmu', bad)
  bad = {k: v for k, v in util.per_leaf_rel_err(model, g_d[1], grho_o).items() if v > 5e-3}
  assert not bad, ('grho', bad)
  eng.close()


def test_vi_minibatch_shares_one_batch():
  n_rows, B = 200, 64
  net, model, X, y = util.make_problem(n_rows=n_rows, width=64, depth=1)
  eng = _engine(net, X, y, mode='vi', members=3, vi_samples=2, kl_weight=0.1, batch=B, seed=9,
                compute_dtype='fp32')
  eng.init_params(0.0)
  p0 = eng.get_params().astype(np.float64)
  r0, r1 = eng.debug_row_index(0, 0), eng.debug_row_index(0, 1)
  assert np.array_equal(r0[0], r0[1]) and np.array_equal(r0[0], r0[2])   # inference.py:704-709
  assert not np.array_equal(r0[0], r1[0]) and len(set(r0[0].tolist())) == B
  eps = eng.debug_vi_eps(1)
  loss_d, g_d = eng.debug_loss_and_grad(0, 1)
  rows = r1[0]
  loss_o, gmu_o, grho_o = O.vi_loss_and_grad(model, p0[0], p0[1], eps, X[rows], y[rows], n_rows, 0.1)
  np.testing.assert_allclose(loss_d, loss_o * 0.1, rtol=5e-5)
  assert max(util.per_leaf_rel_err(model, g_d[0], gmu_o).values()) < 5e-4
  eng.close()


# --------------------------------------------------------------------------- predict
def test_forward_only_and_quantiles():
  from bayesnf_amd.engine import Engine
  net, model, X, y = util.make_problem(n_rows=700, width=64, depth=2)
  M = 11
  theta = util.random_theta(model, M, scale=0.4)
  eng = Engine(net, members=4, forward_only=True, row_capacity=256, compute_dtype='fp32')
  th_d = torch.tensor(theta, dtype=torch.float32, device=eng.device)
  X_d = torch.tensor(X, dtype=torch.float32, device=eng.device)
  loc, aux = eng.forward(th_d, X_d)          # 3 member chunks x 3 row chunks
  torch.cuda.synchronize()
  mu_o, sd_o = O.predict_normal(model, theta, X)
  assert util.rel_err(loc.cpu().numpy(), mu_o) < 2e-4
  np.testing.assert_allclose(aux[:, 0].cpu().numpy(), sd_o, rtol=1e-5)
  qs = (0.5, 0.025, 0.975)
  q_d = eng.normal_mixture_quantiles(loc, aux[:, 0], qs).cpu().numpy()
  for i, q in enumerate(qs):
    np.testing.assert_allclose(O.mixture_cdf(mu_o, sd_o, q_d[i]), q, atol=2e-5)
    np.testing.assert_allclose(q_d[i], O.normal_quantile_via_root(mu_o, sd_o, q), atol=2e-3)
  qa = eng.normal_mixture_quantiles(loc, aux[:, 0], qs, approximate=True).cpu().numpy()
  for i, q in enumerate(qs):
    np.testing.assert_allclose(qa[i], O.approximate_normal_quantile(mu_o, sd_o, q), rtol=1e-4, atol=1e-4)
  eng.close()


# --------------------------------------------------------------------------- count models
@pytest.mark.parametrize('obs', ['NB', 'ZINB'])
@pytest.mark.parametrize('dtype', util.FP32_DTYPES)
def test_count_models_loss_grad_and_training_fp32(obs, dtype):
  """NB / ZINB likelihood (models.py:166-191): step loss, every gradient leaf (incl. `shape`
  and `inflated_loc_probs`), then 100 Adam steps against the oracle (SURVEY 8d's fp32 gates)."""
  n_rows, E = 260, 3
  net, model, X, y = util.make_problem(n_rows=n_rows, width=64, depth=2, observation_model=obs)
  assert (y == 0).sum() > 40 and y.max() >= 3
  theta = util.random_theta(model, E, scale=0.4)
  for pw in (1.0, 0.0):
    eng = _engine(net, X, y, members=E, prior_weight=pw, compute_dtype=dtype)
    eng.set_params(theta)
    loss_d, g_d = eng.debug_loss_and_grad()
    loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows, prior_weight=pw)
    np.testing.assert_allclose(loss_d, loss_o, rtol=G['loss'])
    errs = util.per_leaf_rel_err(model, g_d, g_o)
    bad = {k: v for k, v in errs.items() if v > G['grad']}
    assert not bad, bad
    if pw == 0.0:   # likelihood-only gradient of the unused observation leaves is exactly 0
      assert np.all(g_d[:, model.leaf['log_noise_scale'].offset] == 0)
      if obs == 'NB':
        assert np.all(g_d[:, model.leaf['inflated_loc_probs'].offset] == 0)
    eng.close()
  eng = _engine(net, X, y, members=E, seed=5, learning_rate=0.005, compute_dtype=dtype)
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  theta0 = eng.get_params().astype(np.float64)
  losses = eng.train(0, 100)
  torch.cuda.synchronize()
  theta_o, losses_o = O.train_map(model, theta0, X, y, lr=0.005, num_epochs=100)
  np.testing.assert_allclose(losses.cpu().numpy(), losses_o, rtol=G['loss'])
  assert util.rel_err(eng.get_params(), theta_o) < G['params100'], util.rel_err(eng.get_params(), theta_o)
  eng.close()
  with pytest.raises(ValueError):
    _engine(net, X, y, members=E, compute_dtype='fp32', pipeline='panel')   # bf16 only


@pytest.mark.parametrize('obs', ['NB', 'ZINB'])
def test_count_models_forecast_means_and_quantiles(obs):
  from bayesnf_amd.engine import Engine
  net, model, X, y = util.make_problem(n_rows=333, width=64, depth=2, observation_model=obs)
  M = 7
  theta = util.random_theta(model, M, scale=0.4)
  theta[:, model.leaf['shape'].offset] += np.linspace(-1.0, 1.5, M)
  eng = Engine(net, members=4, forward_only=True, row_capacity=128, compute_dtype='fp32')
  loc, aux = eng.forward(torch.tensor(theta, dtype=torch.float32, device=eng.device),
                         torch.tensor(X, dtype=torch.float32, device=eng.device))
  qs = (0.5, 0.025, 0.975, 0.2)
  means, q_d = eng.count_mixture_quantiles(loc, aux, qs)
  torch.cuda.synchronize()
  fc = O.count_forecast(model, theta, O.forward(model, theta, X))
  assert util.rel_err(means.cpu().numpy(), fc['mean']) < 3e-4
  q_d = q_d.cpu().numpy()
  assert np.all(q_d == np.round(q_d)) and np.all(q_d >= 0)
  for i, q in enumerate(qs):
    q_o = O.count_quantile_via_root(fc, q)
    # the integer quantile is the smallest k with mixture cdf(k) >= q; roots that land within
    # the 1e-5 value tolerance of an integer may round either way
    agree = np.mean(q_d[i] == q_o)
    assert agree > 0.98, (q, agree)
    k = q_d[i]
    assert np.all(O.count_cdf(fc, k[None, :]).mean(axis=0) >= q - 2e-5)
    below = {kk: v for kk, v in fc.items()}
    lo = np.maximum(k - 1, 0)
    F_lo = O.count_cdf(below, lo[None, :]).mean(axis=0)
    assert np.all((F_lo <= q + 2e-5) | (k == 0))
  eng.close()


# --------------------------------------------------------------------------- bf16
@pytest.mark.parametrize('pipeline', ['layers', 'auto'])
def test_bf16_tracks_fp32(pipeline):
  n_rows, E, steps = 512, 4, 40
  net, model, X, y = util.make_problem(n_rows=n_rows, width=128, depth=2)
  res = {}
  for dt in ('fp32', 'bf16'):
    eng = _engine(net, X, y, members=E, seed=2, compute_dtype=dt, pipeline=pipeline)
    eng.init_params(float(np.log(np.nanstd(y) / 2)))
    if dt == 'fp32':
      theta0 = eng.get_params()
    else:
      np.testing.assert_array_equal(eng.get_params(), theta0)   # init does not depend on dtype
    l0, g0 = eng.debug_loss_and_grad()
    losses = eng.train(0, steps).cpu().numpy()
    res[dt] = (l0, g0, losses, eng.get_params())
    eng.close()
  np.testing.assert_allclose(res['bf16'][0], res['fp32'][0], rtol=2e-3)
  assert util.rel_err(res['bf16'][1], res['fp32'][1]) < 0.05
  np.testing.assert_allclose(res['bf16'][2][:, -1], res['fp32'][2][:, -1], rtol=1e-2)
  assert np.all(res['bf16'][2][:, -1] < res['bf16'][2][:, 0])


def test_extreme_quantiles_normal_and_counts():
  """Tail quantiles: the mixture CDF at the returned point must hit q within the root finder's
  value tolerance (Normal), and the integer count quantile must be the smallest k with CDF >= q."""
  from bayesnf_amd.engine import Engine
  for obs in ('NORMAL', 'NB'):
    net, model, X, y = util.make_problem(n_rows=150, width=64, depth=1, observation_model=obs)
    M = 5
    theta = util.random_theta(model, M, scale=0.4)
    eng = Engine(net, members=M, forward_only=True, row_capacity=150, compute_dtype='fp32')
    loc, aux = eng.forward(torch.tensor(theta, dtype=torch.float32, device=eng.device),
                           torch.tensor(X, dtype=torch.float32, device=eng.device))
    qs = (0.001, 0.01, 0.99, 0.999)
    if obs == 'NORMAL':
      mu_o, sd_o = O.predict_normal(model, theta, X)
      q_d = eng.normal_mixture_quantiles(loc, aux[:, 0], qs).cpu().numpy()
      for i, q in enumerate(qs):
        np.testing.assert_allclose(O.mixture_cdf(mu_o, sd_o, q_d[i]), q, atol=2e-5)
    else:
      fc = O.count_forecast(model, theta, O.forward(model, theta, X))
      _, q_d = eng.count_mixture_quantiles(loc, aux, qs)
      q_d = q_d.cpu().numpy()
      for i, q in enumerate(qs):
        k = q_d[i]
        assert np.all(O.count_cdf(fc, k[None, :]).mean(axis=0) >= q - 3e-5)
        lo = np.maximum(k - 1, 0)
        assert np.all((O.count_cdf(fc, lo[None, :]).mean(axis=0) <= q + 3e-5) | (k == 0))
    eng.close()


def test_graph_replayed_training_equals_eager(monkeypatch):
  """BNF_GRAPH=1: full-batch MAP steps replayed from one captured hipGraph (per-step Adam bias
  corrections and loss column read from device memory) give the eager loop's losses and parameters."""
  net, model, X, y = util.make_problem(n_rows=200, width=64, depth=2)
  out = {}
  for mode in ('0', '1'):
    monkeypatch.setenv('BNF_GRAPH', mode)
    eng = _engine(net, X, y, members=3, seed=4, learning_rate=0.005, compute_dtype='fp32')
    eng.init_params(0.1)
    l1 = eng.train(0, 12)
    l2 = eng.train(12, 9)          # second call: new loss tensor, Adam step count carries on
    torch.cuda.synchronize()
    out[mode] = (np.concatenate([l1.cpu().numpy(), l2.cpu().numpy()], axis=1), eng.get_params())
    eng.close()
  np.testing.assert_allclose(out['1'][0], out['0'][0], rtol=1e-5)
  assert util.rel_err(out['1'][1], out['0'][1]) < 1e-5


@pytest.mark.parametrize('dtype,pipeline,width,depth', [('fp32', 'layers', 192, 3), ('fp32', 'auto', 256, 3),
                                                       ('bf16', 'auto', 256, 3), ('bf16', 'panel', 512, 2)])
def test_results_do_not_depend_on_stale_lds(dtype, pipeline, width, depth):
  """LDS is not cleared between kernels: a kernel that read LDS it has not written would see the previous
  kernel's leftovers, i.e. results that depend on what ran before (one candidate for the unexplained
  one-off failure of test_forward_and_grad_fp32 in round 2).  The same loss + gradient evaluation with every
  CU's LDS filled with quiet NaNs, with zeros and with a bit pattern right before it: identical up to the
  order of the f32 atomics, and finite."""
  n_rows, E = 257, 3
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  theta = util.random_theta(model, E, scale=0.3)
  eng = _engine(net, X, y, members=E, compute_dtype=dtype, pipeline=pipeline)
  eng.set_params(theta)
  loss0, g0 = eng.debug_loss_and_grad()
  scale = np.abs(g0).max(axis=1, keepdims=True)
  for pattern in (0x7fc00000, 0x00000000, 0x5a5a5a5a):
    eng.debug_poison_lds(pattern)
    loss, g = eng.debug_loss_and_grad()
    assert np.all(np.isfinite(loss)) and np.all(np.isfinite(g)), hex(pattern)
    np.testing.assert_allclose(loss, loss0, rtol=1e-5)
    assert (np.abs(g - g0) / scale).max() < 1e-4, hex(pattern)
  eng.close()
