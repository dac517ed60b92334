"""The row-panel forward + backward kernel (pipeline 'panel': bf16, depth 2, width 256 / 512;
bayesnf_amd/csrc/bnf_panel.h) against the oracle and against the layer pipeline in the same
arithmetic.  bf16 contractions: statistical bars vs the float64 oracle (loss 5e-3, gradient leaves
6e-2 of the leaf's max); against the bf16 layer pipeline -- same operand rounding, different
accumulation order and an un-rounded recomputed A_0 -- loss 1e-3 and leaves 2e-2."""
import numpy as np
import pytest
import torch

from oracle import bnf_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _engine(net, X, y, **kw):
  from bayesnf_amd.engine import Engine
  return Engine(net, X=X, y=y, **kw)


def _leaf_errs(model, g, ref):
  return util.per_leaf_rel_err(model, g, ref)


@pytest.mark.parametrize('width,n_rows,obs', [(256, 300, 'NORMAL'), (512, 300, 'NORMAL'), (512, 1000, 'NORMAL'),
                                              (256, 700, 'NB'), (512, 260, 'ZINB')])
def test_panel_step_vs_oracle_and_layer_pipeline(width, n_rows, obs):
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=2, observation_model=obs)
  E = 3
  theta = util.random_theta(model, E, scale=0.3)
  res = {}
  for pipe in ('panel', 'layers'):
    eng = _engine(net, X, y, members=E, compute_dtype='bf16', pipeline=pipe)
    eng.set_params(theta)
    res[pipe] = eng.debug_loss_and_grad()
    if pipe == 'panel':
      out_d = eng.debug_activation(200)
      H1 = eng.debug_activation(1)
      dZ1, dZ0 = eng.debug_activation(301), eng.debug_activation(300)
    else:
      H1_l, dZ1_l, dZ0_l = eng.debug_activation(1), eng.debug_activation(301), eng.debug_activation(300)
    eng.close()
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  out_o, ch = O.forward(model, theta, X, keep=True)
  # activations written for the weight-gradient contractions
  assert util.rel_err(H1, ch['Hs'][1]) < 2e-2
  assert util.rel_err(H1, H1_l) < 1e-2
  assert util.rel_err(dZ1, dZ1_l) < 3e-2 and util.rel_err(dZ0, dZ0_l) < 3e-2
  assert util.rel_err(out_d, out_o) < 3e-2
  loss_p, g_p = res['panel']
  loss_l, g_l = res['layers']
  np.testing.assert_allclose(loss_p, loss_o, rtol=5e-3)
  np.testing.assert_allclose(loss_p, loss_l, rtol=1e-3)
  bad = {k: v for k, v in _leaf_errs(model, g_p, g_o).items() if v > 6e-2}
  assert not bad, ('vs oracle', bad)
  bad = {k: v for k, v in _leaf_errs(model, g_p, g_l).items() if v > 2e-2}
  assert not bad, ('vs layers', bad)


def test_panel_minibatch_and_training_tracks_fp32():
  n_rows, B, E = 1500, 600, 4
  net, model, X, y = util.make_problem(n_rows=n_rows, width=256, depth=2)
  out = {}
  for name, kw in [('panel', dict(compute_dtype='bf16', pipeline='panel')), ('fp32', dict(compute_dtype='fp32'))]:
    eng = _engine(net, X, y, members=E, batch=B, seed=5, learning_rate=0.005, **kw)
    eng.init_params(0.3)
    losses = eng.train(0, 6)
    torch.cuda.synchronize()
    out[name] = (losses.cpu().numpy(), eng.get_params())
    eng.close()
  lp, l32 = out['panel'][0], out['fp32'][0]
  assert np.all(np.isfinite(lp)) and np.all(lp[:, -1] < lp[:, 0])
  np.testing.assert_allclose(lp, l32, rtol=2e-2)
  assert np.abs(out['panel'][1] - out['fp32'][1]).max() < 0.05


def test_panel_vi_step():
  n_rows, E, S = 400, 2, 3
  net, model, X, y = util.make_problem(n_rows=n_rows, width=256, depth=2)
  res = {}
  for pipe in ('panel', 'layers'):
    eng = _engine(net, X, y, mode='vi', members=E, vi_samples=S, kl_weight=0.2, seed=3, learning_rate=0.01,
                  compute_dtype='bf16', pipeline=pipe)
    eng.init_params(0.0)
    res[pipe] = eng.debug_loss_and_grad(0, 0)
    eng.close()
  np.testing.assert_allclose(res['panel'][0], res['layers'][0], rtol=2e-3)
  for k in (0, 1):
    bad = {n: v for n, v in _leaf_errs(model, res['panel'][1][k], res['layers'][1][k]).items() if v > 3e-2}
    assert not bad, (k, bad)


@pytest.mark.parametrize('width,depth,S,n_rows', [(256, 2, 1, 300), (512, 2, 7, 260), (1024, 3, 2, 200)])
def test_panel_vi_step_vs_oracle(width, depth, S, n_rows):
  """The VI step of the row-panel pipeline against the float64 oracle on the noise the device drew (bnf_debug_vi_eps): the
  Dense kernels' samples exist only as the bf16 fragments k_vi_sample_pack wrote, k_vi_adam makes the noise again -- loss and
  d mu / d rho at the panel kernel's bf16 bars, for one sample, for a count that is no multiple of four, and for the
  width-1024 form."""
  E = 2
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  eng = _engine(net, X, y, mode='vi', members=E, vi_samples=S, kl_weight=0.2, seed=5, learning_rate=0.01,
                compute_dtype='bf16', pipeline='panel')
  eng.init_params(0.0)
  p0 = eng.get_params().astype(np.float64)
  eps0 = eng.debug_vi_eps(0)
  assert eps0.shape == (E, S, model.P)
  loss_d, g_d = eng.debug_loss_and_grad(0, 0)
  eng.close()
  loss_o, gmu_o, grho_o = O.vi_loss_and_grad(model, p0[0], p0[1], eps0, X, y, n_rows, 0.2)
  np.testing.assert_allclose(loss_d, loss_o * 0.2, rtol=5e-3)
  bad = {k: v for k, v in _leaf_errs(model, g_d[0], gmu_o).items() if v > 6e-2}
  assert not bad, ('gmu', bad)
  bad = {k: v for k, v in _leaf_errs(model, g_d[1], grho_o).items() if v > 6e-2}
  assert not bad, ('grho', bad)


def test_panel_repeated_step_is_reproducible():
  net, model, X, y = util.make_problem(n_rows=2000, width=512, depth=2)
  eng = _engine(net, X, y, members=4, seed=1, compute_dtype='bf16', pipeline='panel')
  eng.init_params(0.2)
  loss0, g0 = eng.debug_loss_and_grad()
  scale = np.abs(g0).max(axis=1, keepdims=True)
  for _ in range(8):
    loss, g = eng.debug_loss_and_grad()
    assert np.abs(loss - loss0).max() <= 1e-5 * np.abs(loss0).max()
    assert (np.abs(g - g0) / scale).max() < 1e-4
  eng.close()


def test_fused_featurisation_backward_matches_the_separate_kernel(monkeypatch):
  """W = 512, <= 64 features: the panel kernel finishes the featurisation backward itself (d feature
  scales, d log_scale_adjustment from the dH0 tiles in registers and the bf16 feature panel in LDS;
  bnf_panel.h) instead of writing dH0^T for k_feat_bwd.  Both against the float64 oracle (2e-2 of the
  leaf's max; measured <= 9e-3 fused, <= 6e-3 separate: scripts/featbwd_diag.py) and each other."""
  n_rows, E = 1000, 3
  net, model, X, y = util.make_problem(n_rows=n_rows, width=512, depth=2)
  theta = util.random_theta(model, E, scale=0.3)
  _, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  g = {}
  for flag in ('1', '0'):
    monkeypatch.setenv('BNF_PANEL_FEATBWD', flag)
    eng = _engine(net, X, y, members=E, compute_dtype='bf16', pipeline='panel')
    eng.set_params(theta)
    g[flag] = eng.debug_loss_and_grad()[1]
    g2 = eng.debug_loss_and_grad()[1]          # LDS group sums are re-zeroed by every workgroup
    assert util.rel_err(g2, g[flag]) < 2e-3
    eng.close()
  names = [k for k in model.leaf if k.startswith('feature_inv_sp_scale') or k == 'log_scale_adjustment']
  assert len(names) >= 5
  for flag in g:
    errs = util.per_leaf_rel_err(model, g[flag], g_o)
    bad = {k: errs[k] for k in names if errs[k] > 2e-2}
    assert not bad, (flag, bad)
  errs = util.per_leaf_rel_err(model, g['1'], g['0'])
  bad = {k: errs[k] for k in names if errs[k] > 1e-2}
  assert not bad, bad
  rest = {k: v for k, v in errs.items() if k not in names and v > 2e-3}   # nothing else changes
  assert not rest, rest


@pytest.mark.parametrize('depth,n_rows,harmonics', [(2, 700, (20, 20)), (2, 130, (20, 20)), (3, 1000, (20, 20)),
                                                     (2, 700, (2, 10)), (4, 300, (2, 10))])
def test_w256_with_up_to_128_features_keeps_the_feature_panel_in_lds(monkeypatch, depth, n_rows, harmonics):
  """W = 256 (65 .. 128 padded features: the C5 shape, 105 features here; <= 64: the C1 shape): 128-row panels of two 64-row
  blocks, the feature panel (128 x 272 bytes) staged in LDS for both layer-0 passes, the wave's sixteen layer-0
  weight fragments in registers, featurisation backward fused (k_panel_fwd_bwd<4, 2, true, ., 1, 128>).
  Against the float64 oracle and the bf16 layer pipeline -- loss, every gradient leaf, H1, dZ of both ends,
  output -- and against the two other routes to the same numbers: the separate featurisation-backward kernel
  (BNF_PANEL_FEATBWD=0) and the 256-row panel kernel without the LDS feature panel (BNF_PANEL_NO_H0L=1)."""
  net, model, X, y = util.make_problem(n_rows=n_rows, width=256, depth=depth, periods=(52.1775, 365.25), harmonics=harmonics)
  assert (64 < net.F <= 128) if harmonics == (20, 20) else net.F <= 64      # FP = 128 / FP = 64 forms of the kernel
  E = 3
  theta = util.random_theta(model, E, scale=0.3)
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  out_o, ch = O.forward(model, theta, X, keep=True)
  res = {}
  for route, env in (('lds', {}), ('featbwd_apart', {'BNF_PANEL_FEATBWD': '0'}), ('no_lds', {'BNF_PANEL_NO_H0L': '1'}),
                     ('layers', {})):
    for k in ('BNF_PANEL_FEATBWD', 'BNF_PANEL_NO_H0L'):
      monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    eng = _engine(net, X, y, members=E, compute_dtype='bf16', pipeline='layers' if route == 'layers' else 'panel')
    eng.set_params(theta)
    loss, g = eng.debug_loss_and_grad()
    loss2, g2 = eng.debug_loss_and_grad()           # LDS scratch re-initialised by every workgroup
    assert util.rel_err(g2, g) < 2e-3
    res[route] = (loss, g, eng.debug_activation(1), eng.debug_activation(300), eng.debug_activation(300 + depth - 1),
                  eng.debug_activation(200))
    eng.close()
  for route in ('lds', 'featbwd_apart', 'no_lds'):
    loss, g, H1, dZ0, dZl, out = res[route]
    np.testing.assert_allclose(loss, loss_o, rtol=5e-3)
    np.testing.assert_allclose(loss, res['layers'][0], rtol=1e-3)
    assert util.rel_err(H1, ch['Hs'][1]) < 2e-2 and util.rel_err(H1, res['layers'][2]) < 1e-2
    assert util.rel_err(dZ0, res['layers'][3]) < 3e-2 and util.rel_err(dZl, res['layers'][4]) < 3e-2
    assert util.rel_err(out, out_o) < 3e-2
    bad = {k: v for k, v in _leaf_errs(model, g, g_o).items() if v > 6e-2}
    assert not bad, (route, 'vs oracle', bad)
    bad = {k: v for k, v in _leaf_errs(model, g, res['layers'][1]).items() if v > 2e-2}
    assert not bad, (route, 'vs layers', bad)
  names = [k for k in model.leaf if k.startswith('feature_inv_sp_scale') or k == 'log_scale_adjustment']
  errs = util.per_leaf_rel_err(model, res['lds'][1], res['featbwd_apart'][1])
  bad = {k: errs[k] for k in names if errs[k] > 1e-2}
  assert not bad, bad
  rest = {k: v for k, v in errs.items() if k not in names and v > 2e-3}   # nothing else changes
  assert not rest, rest


@pytest.mark.parametrize('n_rows', [300, 1000, 2333])
def test_weight_gradient_stream_kernels_match_the_two_stage_kernel(n_rows, monkeypatch):
  """gemm_tn_skinny (layer 0 as a 64 x 512 row stream) and gemm_tn_ring (256 x 256 tile, four-stage
  ring, software-pipelined transpose reads) against gemm_tn's two-stage loop on the same activations:
  the same bf16 products, only the f32 summation order differs (split-K, stage size).  Row counts
  give K loops of 10 / 32 / 74 stages: ring fill, drain and the not-a-multiple-of-four tail."""
  net, model, X, y = util.make_problem(n_rows=n_rows, width=512, depth=2)
  E = 3
  theta = util.random_theta(model, E, scale=0.3)
  grads = {}
  for name, env in (('stream', {}), ('two_stage', {'BNF_WGRAD_SKINNY': '0', 'BNF_TN_RING': '0'})):
    for k in ('BNF_WGRAD_SKINNY', 'BNF_TN_RING'):
      monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
      monkeypatch.setenv(k, v)    # read at bnf_create
    eng = _engine(net, X, y, members=E, compute_dtype='bf16', pipeline='panel')
    eng.set_params(theta)
    grads[name] = eng.debug_loss_and_grad()[1]
    eng.close()
  errs = _leaf_errs(model, grads['stream'], grads['two_stage'])
  bad = {k: v for k, v in errs.items() if v > 2e-4}
  assert not bad, bad


@pytest.mark.parametrize('width,depth,n_rows,obs', [(512, 3, 300, 'NORMAL'), (512, 4, 1000, 'NORMAL'), (256, 4, 700, 'NORMAL'),
                                                    (256, 3, 300, 'NB'), (512, 5, 260, 'ZINB'),
                                                    # width 1024: 64-row panels, two 64-column slabs per wave (C4's shape class)
                                                    (1024, 2, 300, 'NORMAL'), (1024, 4, 700, 'NORMAL'), (1024, 3, 200, 'ZINB')])
def test_deep_panel_step_vs_oracle_and_layer_pipeline(width, depth, n_rows, obs):
  """Depth > 2 through the row-panel kernel (round 3: the middle layers' contractions run out of the same LDS
  panel, their pre-activations are parked in HBM in the owning wave's register order and read back by the same
  lanes in the backward pass -- bnf_panel.h) against the float64 oracle and against the bf16 layer pipeline:
  loss, every gradient leaf, the activations H_{l+1} and dZ_l of EVERY layer as the weight-gradient kernels
  read them, the network output."""
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth, observation_model=obs)
  E = 3
  theta = util.random_theta(model, E, scale=0.3)
  res, acts = {}, {}
  for pipe in ('panel', 'layers'):
    eng = _engine(net, X, y, members=E, compute_dtype='bf16', pipeline=pipe)
    eng.set_params(theta)
    res[pipe] = eng.debug_loss_and_grad()
    acts[pipe] = dict(out=eng.debug_activation(200),
                      H=[eng.debug_activation(1 + l) for l in range(depth - 1)],
                      dZ=[eng.debug_activation(300 + l) for l in range(depth)])
    eng.close()
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  out_o, ch = O.forward(model, theta, X, keep=True)
  for l in range(depth - 1):
    assert util.rel_err(acts['panel']['H'][l], ch['Hs'][l + 1]) < 2e-2 * (l + 1), l
    assert util.rel_err(acts['panel']['H'][l], acts['layers']['H'][l]) < 1e-2 * (l + 1), l
  for l in range(depth):
    assert util.rel_err(acts['panel']['dZ'][l], acts['layers']['dZ'][l]) < 4e-2, l
  assert util.rel_err(acts['panel']['out'], out_o) < 4e-2
  loss_p, g_p = res['panel']
  loss_l, g_l = res['layers']
  np.testing.assert_allclose(loss_p, loss_o, rtol=5e-3)
  np.testing.assert_allclose(loss_p, loss_l, rtol=1e-3)
  bad = {k: v for k, v in _leaf_errs(model, g_p, g_o).items() if v > 6e-2}
  assert not bad, ('vs oracle', bad)
  bad = {k: v for k, v in _leaf_errs(model, g_p, g_l).items() if v > 3e-2}
  assert not bad, ('vs layers', bad)


def test_deep_panel_vi_step_and_training():
  """C3's shape class: mean-field VI, depth 4, minibatch -- panel vs layer pipeline on one step (loss, d mu, d rho),
  then a few steps of training through both, and the pipeline the engine picks by default IS the panel one."""
  n_rows, E, S, B = 900, 2, 3, 400
  net, model, X, y = util.make_problem(n_rows=n_rows, width=512, depth=4)
  res = {}
  for pipe in ('panel', 'layers', 'auto'):
    eng = _engine(net, X, y, mode='vi', members=E, vi_samples=S, kl_weight=0.2, seed=3, learning_rate=0.01,
                  batch=B, compute_dtype='bf16', pipeline=pipe)
    eng.init_params(0.0)
    step = eng.debug_loss_and_grad(0, 0)
    losses = eng.train(0, 6)
    torch.cuda.synchronize()
    res[pipe] = (step, losses.cpu().numpy(), eng.get_params())
    eng.close()
  np.testing.assert_allclose(res['panel'][0][0], res['layers'][0][0], rtol=2e-3)
  for k in (0, 1):
    bad = {n: v for n, v in _leaf_errs(model, res['panel'][0][1][k], res['layers'][0][1][k]).items() if v > 4e-2}
    assert not bad, (k, bad)
  np.testing.assert_allclose(res['panel'][1], res['layers'][1], rtol=5e-3)
  assert np.all(np.isfinite(res['panel'][2])) and np.abs(res['panel'][2] - res['layers'][2]).max() < 0.05
  np.testing.assert_allclose(res['auto'][1], res['panel'][1], rtol=1e-4)      # auto = panel for this shape


def test_width_1024_panel_without_the_lds_feature_panel(monkeypatch):
  """The width-1024 variant also exists without the staged feature panel / fused featurisation backward
  (BNF_PANEL_NO_H0L, what a feature count above 64 selects): same gradients as with it."""
  n_rows, E = 300, 2
  net, model, X, y = util.make_problem(n_rows=n_rows, width=1024, depth=3)
  theta = util.random_theta(model, E, scale=0.3)
  g = {}
  for flag in ('0', '1'):
    if flag == '1':
      monkeypatch.setenv('BNF_PANEL_NO_H0L', '1')
    eng = _engine(net, X, y, members=E, compute_dtype='bf16', pipeline='panel')
    eng.set_params(theta)
    g[flag] = eng.debug_loss_and_grad()
    eng.close()
  np.testing.assert_allclose(g['0'][0], g['1'][0], rtol=1e-4)
  bad = {k: v for k, v in _leaf_errs(model, g['0'][1], g['1'][1]).items() if v > 1.5e-2}
  assert not bad, bad


@pytest.mark.parametrize('degrees,inter,want_f', [((6, 5, 5), ((0, 1), (1, 2), (0, 2)), 62),
                                                  ((7, 5, 5), ((0, 1), (1, 2)), 63),
                                                  ((7, 5, 5), ((0, 1), (1, 2), (0, 2)), 64)])
def test_layer0_fold_boundary_feature_counts(degrees, inter, want_f, monkeypatch):
  """Round 4: the H0L forms fold layer 0's scale and bias into its contraction (two ones columns behind the features,
  bias rows in the forward-packed weights, d bias0 = the ones row of the layer-0 weight gradient, swapped operand
  roles in both layer-0 epilogues) -- possible only while F + 2 <= Fp.  F = 62 is the last folded count at Fp = 64,
  63 and 64 run the unfolded form; all three against the float64 oracle, and the folded form against the unfolded
  one (`BNF_PANEL_FOLD0=0`) in the same arithmetic."""
  n_rows, E = 700, 3
  net, model, X, y = util.make_problem(n_rows=n_rows, width=512, depth=2, fourier_degrees=degrees, interactions=inter)
  assert net.F == want_f
  theta = util.random_theta(model, E, scale=0.3)
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  res = {}
  for fold in ('1', '0'):
    monkeypatch.setenv('BNF_PANEL_FOLD0', fold)
    eng = _engine(net, X, y, members=E, compute_dtype='bf16', pipeline='panel')
    eng.set_params(theta)
    res[fold] = eng.debug_loss_and_grad() + (eng.debug_activation(1), eng.debug_activation(300))
    eng.close()
  for fold, (loss_p, g_p, H1, dZ0) in res.items():
    np.testing.assert_allclose(loss_p, loss_o, rtol=5e-3)
    bad = {k: v for k, v in _leaf_errs(model, g_p, g_o).items() if v > 6e-2}
    assert not bad, (fold, 'vs oracle', bad)
  bad = {k: v for k, v in _leaf_errs(model, res['1'][1], res['0'][1]).items() if v > 2e-2}
  assert not bad, ('folded vs unfolded', bad)
  assert util.rel_err(res['1'][2], res['0'][2]) < 1e-2 and util.rel_err(res['1'][3], res['0'][3]) < 3e-2


@pytest.mark.parametrize('width,degrees,mode,batch', [(512, (5, 3, 2), 'map', None), (512, (5, 3, 2), 'map', 300),
                                                      (256, (5, 3, 2), 'map', 256), (1024, (5, 3, 2), 'map', None),
                                                      (256, (12, 9, 9), 'map', None), (512, (5, 3, 2), 'vi', 200)])
def test_panel_featurises_its_own_rows_like_k_featurize(width, degrees, mode, batch, monkeypatch):
  """Round 4 experiment (`-DBNF_PANEL_FIN=1` builds, `scripts/build_variant.sh fin -DBNF_PANEL_FIN=1`, run
  with `BNF_LIB=ab/libbnf_fin.so`): the H0L forms of the panel kernel featurise their rows themselves (no k_featurize launch,
  no re-read of H0) and write the row-major copy the layer-0 weight gradient reads.  `BNF_PANEL_FIN=0` runs the separate
  kernel: the features (`debug_activation(0)`), the loss and every gradient leaf must come out the same -- full batch, the
  engine's shuffled minibatches (row source modes 1 / 2), 64 and 128 padded features, all three widths, MAP and VI.  Green on
  the experiment build (profiles/r04_panel_ab.md r04l); it measured SLOWER than the separate kernel, so the default build
  compiles the path out and this test then compares the default path with itself."""
  n_rows, E = 700, 3
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=2, fourier_degrees=degrees)
  out = {}
  for fin in ('1', '0'):
    monkeypatch.setenv('BNF_PANEL_FIN', fin)
    kw = dict(mode='vi', vi_samples=2, kl_weight=0.2) if mode == 'vi' else {}
    eng = _engine(net, X, y, members=E, compute_dtype='bf16', pipeline='panel', batch=batch, seed=11, **kw)
    eng.init_params(0.2)
    loss, g = eng.debug_loss_and_grad(0, 1 if batch else 0)
    H0 = eng.debug_activation(0)
    losses = eng.train(0, 2).cpu().numpy()
    out[fin] = (loss, g, H0, losses, eng.get_params())
    eng.close()
  a, b = out['1'], out['0']
  np.testing.assert_array_equal(a[2], b[2])                       # the same features, bit for bit
  np.testing.assert_allclose(a[0], b[0], rtol=1e-5)
  ga, gb = np.asarray(a[1]), np.asarray(b[1])
  assert np.max(np.abs(ga - gb)) <= 1e-5 * max(1.0, np.max(np.abs(gb)))    # (f32 atomics order only)
  np.testing.assert_allclose(a[3], b[3], rtol=1e-5)
  assert np.max(np.abs(a[4] - b[4])) < 1e-4


def test_vi_sampler_data_flows_agree(monkeypatch):
  """Round 4: with the device generator the VI step samples the Dense kernels in the kernel that packs them
  (k_vi_sample_pack: the f32 samples of the kernels never reach memory) and k_vi_adam makes the noise again from the
  counter-based stream.  Against the two-kernel flow (BNF_VI_SAMPLE_PACK=0: k_vi_sample writes every sample, k_pack_layers
  packs them) the fused one is the SAME arithmetic on the same normals -- identical losses and parameters; against the
  round-3 flow (BNF_VI_KEEP_Z=1: noise recovered as (z - mu) / sigma from the stored samples) it differs by the rounding
  of z only.  The noise the tests read back (bnf_debug_vi_eps) is the noise both flows draw: one step's d mu equals the
  mean over samples of the likelihood + prior gradient only if eps is the stream's."""
  n_rows, E, S, B = 700, 3, 5, 300
  net, model, X, y = util.make_problem(n_rows=n_rows, width=512, depth=3)
  res = {}
  for name, env in (('fused', {}), ('two_kernels', {'BNF_VI_SAMPLE_PACK': '0'}), ('keep_z', {'BNF_VI_KEEP_Z': '1'}),
                    ('round3', {'BNF_VI_SAMPLE_PACK': '0', 'BNF_VI_KEEP_Z': '1'})):
    for k in ('BNF_VI_SAMPLE_PACK', 'BNF_VI_KEEP_Z'):
      monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    eng = _engine(net, X, y, mode='vi', members=E, vi_samples=S, kl_weight=0.2, seed=11, learning_rate=0.01,
                  batch=B, compute_dtype='bf16', pipeline='panel')
    eng.init_params(0.0)
    eps = eng.debug_vi_eps(0)
    step = eng.debug_loss_and_grad(0, 0)
    losses = eng.train(0, 5)
    torch.cuda.synchronize()
    res[name] = (eps, step, losses.cpu().numpy(), eng.get_params())
    eng.close()
  f = res['fused']
  assert f[0].shape == (E, S, model.P) and abs(float(f[0].std()) - 1.0) < 5e-3 and abs(float(f[0].mean())) < 5e-3
  for other in ('two_kernels', 'keep_z', 'round3'):
    o = res[other]
    np.testing.assert_array_equal(f[0], o[0])                      # one stream
    same = other == 'two_kernels'      # same arithmetic; only the order of the gradient atomics differs from run to run
    np.testing.assert_allclose(f[1][0], o[1][0], rtol=1e-6 if same else 1e-5)
    for k in (0, 1):                                               # d mu, d rho of the first step
      np.testing.assert_allclose(f[1][1][k], o[1][1][k], rtol=0, atol=(2e-5 if same else 5e-4) * np.abs(o[1][1][k]).max())
    np.testing.assert_allclose(f[2], o[2], rtol=1e-5 if same else 1e-4)
    assert np.abs(f[3] - o[3]).max() <= (1e-3 if same else 5e-3)      # five Adam steps of 0.01 on top of that


@pytest.mark.parametrize('width,depth,n_rows', [(512, 2, 1000), (256, 2, 700), (512, 3, 600)])
def test_bias_and_output_kernel_gradients_keep_explicit_bars(width, depth, n_rows, monkeypatch):
  """Leaves the row-panel kernel sums over bf16-ROUNDED panel values (f32 accumulation on the matrix pipe) instead of over
  its f32 registers: `Dense_0/bias` since round 4 (the ones row of the layer-0 weight gradient, F0 forms; the forward bias is
  a bf16 hi + lo pair), `Dense_L/bias` (1^T dZ_L) and the output kernel (dv^T H_{L+1}) since round 5 (L1T forms).  Explicit
  bars on exactly those leaves, against the float64 oracle and against the fp32 engine: 6e-3 of the leaf's max (measured
  <= 2.5e-3, scripts/l1t_leaf_diag.py: gpurun_out/r05a; the f32-register sums of round 3 / 4 measured 3e-4 .. 1.6e-3), and
  the folded layer 0 against the unfolded one (`BNF_PANEL_FOLD0=0`: d bias0 from f32 column sums of z) at 4e-3."""
  E = 3
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  theta = util.random_theta(model, E, scale=0.3)
  _, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows)
  res = {}
  for name, kw, fold in (('panel', dict(compute_dtype='bf16', pipeline='panel'), '1'),
                         ('unfolded', dict(compute_dtype='bf16', pipeline='panel'), '0'), ('fp32', dict(compute_dtype='fp32'), '1')):
    monkeypatch.setenv('BNF_PANEL_FOLD0', fold)
    eng = _engine(net, X, y, members=E, **kw)
    eng.set_params(theta)
    res[name] = eng.debug_loss_and_grad()[1]
    eng.close()
  leaves = [f'Dense_{l}/bias' for l in range(depth)] + [f'Dense_{depth}/kernel']
  for ref_name, ref, bar in (('oracle', g_o, 6e-3), ('fp32', res['fp32'], 6e-3), ('unfolded', res['unfolded'], 4e-3)):
    errs = _leaf_errs(model, res['panel'], ref)
    bad = {k: errs[k] for k in leaves if errs[k] > bar}
    assert not bad, (ref_name, bad)
