"""CPU oracle: a numpy restatement of the BayesNF ensemble-training hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
Nothing under `bayesnf_amd/` imports, calls or links anything in `oracle/`.

What it restates (all citations relative to /root/reference):

  * featurisers                 src/bayesnf/models.py:36-88
  * network forward             src/bayesnf/models.py:212-273
  * NORMAL / NB / ZINB likelihood src/bayesnf/models.py:157-191
  * Logistic prior              src/bayesnf/models.py:91-103
  * MAP / MLE loss + loop       src/bayesnf/inference.py:510-623 (scaling :558-569,
                                Adam :580,605-606, batching :583-597, epoch mean :614)
  * MAP / VI initial values     src/bayesnf/inference.py:399-427, :203-231
  * mean-field VI (ELBO)        src/bayesnf/inference.py:626-764
  * predict + quantiles         src/bayesnf/inference.py:42-100, :103-200, :461-507

The arithmetic of jax / flax / optax / tensorflow_probability is NOT under
/root/reference (un-vendored pip dependencies: jax==0.4.26, flax==0.8.3,
optax==0.2.2, tensorflow-probability==0.24.0, see
requirements.Python3.10.14.txt:13,19,20,31,51) and none of them is installable
here, so their published semantics are restated:

  nn.Dense           y = x @ kernel + bias, kernel is (in, out)
  nn.elu             x > 0 ? x : expm1(x)
  optax.adam(lr)     b1=.9 b2=.999 eps=1e-8 eps_root=0, bias-corrected
  tfd.Logistic(m,1)  log_prob(z) = -(z-m) - 2 softplus(-(z-m))
  tfd.Normal         log_prob = -((y-mu)/s)^2/2 - log s - log(2 pi)/2
  JDC.log_prob       sum over every element of every part (batch_ndims=0)
  fit_surrogate_posterior_stateless
                     reparameterised reverse-KL Monte-Carlo loss averaged over
                     `sample_size`; trace = loss before each update
  find_root_chandrupatla   Chandrupatla (1997) bracketing root finder

PARITY PIN STATUS: PINNED.  The reference's golden files for this path
(tests/test_data/bnf-{map,mle,vi}.chickenpox.8.mini.pred.csv, produced by the
three *skipped* tests tests/test_evaluate_mini.py:58-91) are bit-exact functions
of JAX threefry keys.  With the random streams restated in oracle/jax_rng.py
(threefry2x32, jax.random split / fold_in / normal / truncated_normal, TFP's
JointDistribution / vectorised-sample / minimize seed plumbing -- determined
against these very files) this module reproduces column `yhat` of all three
element-wise: MAP 1.9e-6, MLE 5.0e-6, VI 2.5e-6 max abs (tests/test_jax_rng.py),
the quantile columns as roots of the mixture CDF within the reference's own
tolerance.  RNG-independent projections: tests/test_oracle_kat.py (KAT K1-K4 of
SURVEY.md section 8c); gradients additionally against finite differences and
torch autograd (tests/test_oracle_grad.py); NB / ZINB closed forms against
scipy.stats.nbinom (tests/test_oracle_counts.py; the reference has no count golden).

The oracle is RNG-free: initial parameters, minibatch row indices and VI
noise are *inputs* (the tests read them back from the device library).

Everything is batched over a leading member axis E.
"""

from __future__ import annotations

import dataclasses
import math
from typing import Sequence

import numpy as np
from scipy import special as _sp

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------
# featuriser tables                                         models.py:36-59
# --------------------------------------------------------------------------
def make_seasonal_frequencies(seasonality_periods, num_harmonics):
  """Unique h/p frequencies in float32, first-occurrence order (models.py:36-59)."""
  periods = np.array(seasonality_periods, dtype=np.float32)
  num_harmonics = np.asarray(num_harmonics)
  if np.any(num_harmonics > periods / 2):
    raise ValueError('Harmonic cannot exceed half seasonal period.')
  if periods.shape != num_harmonics.shape:
    raise ValueError('Number of seasonal periods and harmonics must be equal.')
  if num_harmonics.ndim != 1:
    raise ValueError(
        'Arguments `num_harmonics` and `seasonality_periods` must be rank 1.')
  if periods.shape[0] == 0:
    return np.zeros(0, np.float32), np.zeros(0, np.float32)
  harmonics = [np.arange(1, h + 1, dtype=np.float32) for h in num_harmonics]
  freqs = np.concatenate([h / p for h, p in zip(harmonics, periods)])
  _, idx = np.unique(freqs, return_index=True)
  idx = np.sort(idx)
  return freqs[idx].astype(np.float32), np.concatenate(harmonics)[idx].astype(
      np.float32)


# --------------------------------------------------------------------------
# model description + parameter layout      models.py:197-273, 94-103
# --------------------------------------------------------------------------
@dataclasses.dataclass
class Leaf:
  name: str
  shape: tuple
  offset: int
  size: int
  is_matrix: bool
  prior_loc: float = 0.0


class Model:
  """Static description of one BayesNF network (what flax's module holds)."""

  def __init__(self, width, depth, input_scales, fourier_degrees,
               interactions, seasonality_periods=(), num_seasonal_harmonics=(),
               observation_model='NORMAL'):
    self.width = int(width)
    self.depth = int(depth)
    self.input_scales = np.asarray(input_scales, dtype=np.float64)
    self.D = int(self.input_scales.shape[0])
    self.fourier_degrees = np.asarray(fourier_degrees, dtype=int)
    assert self.fourier_degrees.shape == (self.D,)
    self.interactions = np.asarray(interactions, dtype=int).reshape(-1, 2)
    self.freqs, self.harm = make_seasonal_frequencies(
        np.asarray(seasonality_periods, dtype=float),
        np.asarray(num_seasonal_harmonics))
    self.observation_model = observation_model

    # feature groups, in the order of models.py:242-247; the scale parameter
    # is named by the index in the UNFILTERED list (models.py:248-251).
    groups = [('u', -1, self.D)]
    for d, deg in enumerate(self.fourier_degrees):
      if deg > 0:
        groups.append(('fourier', d, 2 * int(deg)))
    groups.append(('seasonal', -1, 2 * len(self.freqs)))
    groups.append(('inter', -1, len(self.interactions)))
    self.groups = []  # (kind, arg, ncols, col0, scale_leaf_name)
    col = 0
    for i, (kind, arg, ncols) in enumerate(groups):
      if ncols > 0:
        self.groups.append((kind, arg, ncols, col, f'feature_inv_sp_scale{i}'))
        col += ncols
    self.F = col

    # leaves: [lns, shape, infl] + tree_leaves(flax dict) = string-sorted keys
    mlp = {}
    n_in = self.F
    for l in range(self.depth):
      mlp[f'Dense_{l}'] = (('bias', (self.width,)), ('kernel', (n_in, self.width)))
      n_in = self.width
      mlp[f'inv_sp_layer_scale{l}'] = ()
    mlp[f'Dense_{self.depth}'] = (('bias', (1,)), ('kernel', (self.width, 1)))
    for g in self.groups:
      mlp[g[4]] = ()
    mlp['inv_sp_output_scale'] = ()
    mlp['log_scale_adjustment'] = (self.D,)
    mlp['logit_activation_weight'] = ()
    names_shapes = [('log_noise_scale', ()), ('shape', ()),
                    ('inflated_loc_probs', ())]
    for key in sorted(mlp):
      val = mlp[key]
      if key.startswith('Dense_'):
        for sub, shp in val:
          names_shapes.append((f'{key}/{sub}', shp))
      else:
        names_shapes.append((key, val))
    self.leaves = []
    off = 0
    for name, shp in names_shapes:
      size = int(np.prod(shp)) if shp else 1
      self.leaves.append(Leaf(name, tuple(shp), off, size, len(shp) == 2,
                              -1.5 if name == 'shape' else 0.0))
      off += size
    self.P = off
    self.leaf = {lf.name: lf for lf in self.leaves}

  # -- flat (E, P) <-> dict of (E, *shape) views --------------------------
  def view(self, theta, name):
    lf = self.leaf[name]
    return theta[..., lf.offset:lf.offset + lf.size].reshape(
        theta.shape[:-1] + lf.shape)

  def unpack(self, theta):
    return [self.view(theta, lf.name) for lf in self.leaves]

  def pack(self, leaves_list, dtype=np.float64):
    lead = leaves_list[0].shape[:leaves_list[0].ndim - len(self.leaves[0].shape)]
    out = np.zeros(lead + (self.P,), dtype=dtype)
    for lf, arr in zip(self.leaves, leaves_list):
      out[..., lf.offset:lf.offset + lf.size] = np.asarray(arr).reshape(
          lead + (lf.size,))
    return out

  def prior_loc(self):
    loc = np.zeros(self.P)
    loc[self.leaf['shape'].offset] = -1.5
    return loc

  def matrix_mask(self):
    m = np.zeros(self.P, dtype=bool)
    for lf in self.leaves:
      if lf.is_matrix:
        m[lf.offset:lf.offset + lf.size] = True
    return m


# --------------------------------------------------------------------------
# small math helpers
# --------------------------------------------------------------------------
def softplus(x):
  return np.logaddexp(x, 0.0)


def sigmoid(x):
  return _sp.expit(x)


def _elu(a):
  return np.where(a > 0, a, np.expm1(np.minimum(a, 0)))


def _trig_args(model, x, u, dtype, f32_trig_args):
  """Arguments of the cos/sin features, with the reference's float32 rounding.

  Fourier (models.py:84-85):  y = fl32(fl32(2 pi) * 2^k * u)
  Seasonal (models.py:73):    y = fl32(fl32(fl32(2 pi) * f) * t)
  With f32_trig_args=False the arguments are computed in `dtype` exactly.
  """
  two_pi32 = np.float32(2.0 * np.pi)
  t = x[..., 0]
  fargs = {}
  if f32_trig_args:
    for d, deg in enumerate(model.fourier_degrees):
      if deg > 0:
        c = two_pi32 * np.float32(2.0)**np.arange(deg, dtype=np.float32)
        fargs[d] = (c * u[..., d, None].astype(np.float32)).astype(dtype)
    c_s = (two_pi32 * model.freqs.astype(np.float32)).astype(np.float32)
    sarg = (c_s * t[..., None].astype(np.float32)).astype(dtype)
  else:
    for d, deg in enumerate(model.fourier_degrees):
      if deg > 0:
        c = (2.0 * np.pi * 2.0**np.arange(deg)).astype(dtype)
        fargs[d] = c * u[..., d, None]
    sarg = (2.0 * np.pi * model.freqs.astype(np.float64)).astype(dtype) * t[
        ..., None]
  return fargs, sarg


# --------------------------------------------------------------------------
# forward                                              models.py:212-273
# --------------------------------------------------------------------------
def forward(model: Model, theta, x, dtype=np.float64, f32_trig_args=True,
            keep=False):
  """theta (E,P); x (B,D) shared or (E,B,D) per member -> out (E,B).

  Returns (out, cache) when keep=True.
  """
  theta = np.asarray(theta, dtype=dtype)
  x = np.asarray(x, dtype=dtype)
  E = theta.shape[0]
  if x.ndim == 2:
    x = np.broadcast_to(x, (E,) + x.shape)
  W, L = model.width, model.depth
  lsa = model.view(theta, 'log_scale_adjustment')                 # (E,D)
  s = model.input_scales.astype(dtype) * np.exp(lsa)              # (E,D)
  u = x / s[:, None, :]                                           # (E,B,D)
  fargs, sarg = _trig_args(model, x, u, dtype, f32_trig_args)

  G = []   # unscaled feature groups
  for kind, arg, ncols, col0, sname in model.groups:
    if kind == 'u':
      g = u
    elif kind == 'fourier':
      deg = ncols // 2
      den = np.arange(1, deg + 1).astype(dtype)
      a = fargs[arg]
      g = np.concatenate([np.cos(a) / den, np.sin(a) / den], axis=-1)
    elif kind == 'seasonal':
      den = model.harm.astype(dtype)
      g = np.concatenate([np.cos(sarg) / den, np.sin(sarg) / den], axis=-1)
    else:
      p, q = model.interactions[:, 0], model.interactions[:, 1]
      g = u[..., p] * u[..., q]
    G.append(g.astype(dtype))
  H0 = np.concatenate([
      g * softplus(model.view(theta, grp[4]))[:, None, None]
      for g, grp in zip(G, model.groups)
  ], axis=-1)                                                      # (E,B,F)

  alpha = sigmoid(model.view(theta, 'logit_activation_weight'))[:, None, None]
  Hs, As, gammas = [H0], [], []
  h = H0
  for l in range(L):
    n = h.shape[-1]
    K = model.view(theta, f'Dense_{l}/kernel')
    b = model.view(theta, f'Dense_{l}/bias')
    gam = softplus(model.view(theta, f'inv_sp_layer_scale{l}'))[:, None, None]
    z = np.matmul(h / np.sqrt(dtype(n)), K) + b[:, None, :]
    a = gam * z
    h = alpha * _elu(a) + (1 - alpha) * np.tanh(a)
    As.append(a)
    gammas.append(gam)
    Hs.append(h)
  ko = model.view(theta, f'Dense_{L}/kernel')[..., 0]              # (E,W)
  bo = model.view(theta, f'Dense_{L}/bias')[..., 0]                # (E,)
  gam_o = softplus(model.view(theta, 'inv_sp_output_scale'))       # (E,)
  v = np.einsum('ebw,ew->eb', h / np.sqrt(dtype(W)), ko) + bo[:, None]
  out = gam_o[:, None] * v
  if not keep:
    return out
  cache = dict(x=x, u=u, G=G, Hs=Hs, As=As, gammas=gammas, alpha=alpha, v=v,
               gam_o=gam_o, fargs=fargs)
  return out, cache


# --------------------------------------------------------------------------
# likelihood + prior                    models.py:157-191, 94-103
# --------------------------------------------------------------------------
def noise_scale(model, theta):
  return 0.01 + np.exp(model.view(theta, 'log_noise_scale'))


def normal_loglik(out, y, sigma):
  r = (y - out) / sigma[:, None]
  return np.sum(-0.5 * r * r - np.log(sigma)[:, None] - 0.5 * LOG_2PI, axis=-1)


def log_prior(model, theta):
  z = theta - model.prior_loc().astype(theta.dtype)
  return np.sum(-z - 2.0 * np.logaddexp(-z, 0.0), axis=-1)


def nb_logits_total_count(model, theta, out):
  """models.py:166-176.  Follows the code, not the docstring."""
  mean = np.logaddexp(out, 0.0)
  shape = np.logaddexp(model.view(theta, 'shape'), 0.0)          # (E,)
  total_count = 1.0 / shape
  logits = -np.log(shape)[:, None] - np.log(mean)
  return total_count, logits


def nb_log_prob(y, total_count, logits):
  """tfd.NegativeBinomial(total_count, logits).log_prob (TFP 0.24):

     log_unnorm = tc * log_sigmoid(-logits) + y * log_sigmoid(logits)
     log_norm   = -lgamma(tc + y) + lgamma(1 + y) + lgamma(tc)
  """
  tc = total_count[:, None]
  log_unnorm = tc * (-np.logaddexp(logits, 0.0)) + y * (
      -np.logaddexp(-logits, 0.0))
  log_norm = -_sp.gammaln(tc + y) + _sp.gammaln(1.0 + y) + _sp.gammaln(tc)
  return log_unnorm - log_norm


def zinb_log_prob(y, total_count, logits, pi):
  """tfd.ZeroInflatedNegativeBinomial = Mixture(cat=[1-pi, pi], [NB, delta_0])."""
  lp_nb = nb_log_prob(y, total_count, logits)
  pi = np.broadcast_to(pi, lp_nb.shape)
  at0 = np.logaddexp(np.log1p(-pi) + lp_nb, np.log(pi))
  return np.where(y == 0, at0, np.log1p(-pi) + lp_nb)


def loglik(model, theta, out, y):
  om = model.observation_model
  if om == 'NORMAL':
    return normal_loglik(out, y, noise_scale(model, theta))
  tc, logits = nb_logits_total_count(model, theta, out)
  if om == 'NB':
    return np.sum(nb_log_prob(y, tc, logits), axis=-1)
  if om == 'ZINB':
    pi = sigmoid(model.view(theta, 'inflated_loc_probs'))[:, None]
    return np.sum(zinb_log_prob(y, tc, logits, pi), axis=-1)
  raise AssertionError(om)


# --------------------------------------------------------------------------
# loss + hand-derived gradient              inference.py:558-569 (+autodiff)
# --------------------------------------------------------------------------
def map_loss(model, theta, x, y, n_total, prior_weight=1.0, dtype=np.float64,
             f32_trig_args=True):
  theta = np.asarray(theta, dtype=dtype)
  y = np.asarray(y, dtype=dtype)
  if y.ndim == 1:
    y = np.broadcast_to(y, (theta.shape[0],) + y.shape)
  out = forward(model, theta, x, dtype, f32_trig_args)
  c = dtype(n_total) / dtype(y.shape[-1])
  val = loglik(model, theta, out, y) * c
  if prior_weight != 0.0:
    val = val + log_prior(model, theta) * dtype(prior_weight)
  return -val


def _dloglik_dout_and_params(model, theta, out, y):
  """d loglik / d out (E,B) and direct grads wrt lns / shape / infl (E,)."""
  E = theta.shape[0]
  g_direct = np.zeros_like(theta)
  om = model.observation_model
  if om == 'NORMAL':
    lns = model.view(theta, 'log_noise_scale')
    sigma = 0.01 + np.exp(lns)
    r = y - out
    dout = r / (sigma**2)[:, None]
    dsig = np.sum(r * r, axis=-1) / sigma**3 - y.shape[-1] / sigma
    g_direct[:, model.leaf['log_noise_scale'].offset] = dsig * np.exp(lns)
    return dout, g_direct
  # NB / ZINB:  mean = softplus(out), shape = softplus(th_shape)
  th_shape = model.view(theta, 'shape')
  shape = np.logaddexp(th_shape, 0.0)
  tc = (1.0 / shape)[:, None]
  mean = np.logaddexp(out, 0.0)
  logits = -np.log(shape)[:, None] - np.log(mean)
  sg = sigmoid(logits)
  # d lp_nb / d logits = y * (1 - sg) - tc * sg
  dl_dlogits = y * (1.0 - sg) - tc * sg
  # d lp_nb / d tc = log_sigmoid(-logits) + digamma(tc+y) - digamma(tc)
  dl_dtc = -np.logaddexp(logits, 0.0) + _sp.digamma(tc + y) - _sp.digamma(tc)
  if om == 'ZINB':
    th_pi = model.view(theta, 'inflated_loc_probs')
    pi = sigmoid(th_pi)[:, None]
    lp_nb = nb_log_prob(y, tc[:, 0], logits)
    # weight of the NB branch in d/d(lp_nb):  y>0: 1 ;  y==0: (1-pi)p/((1-pi)p+pi)
    p0 = np.exp(lp_nb)
    w = np.where(y == 0, (1 - pi) * p0 / ((1 - pi) * p0 + pi), 1.0)
    dlp_dpi = np.where(y == 0, (1.0 - p0) / ((1 - pi) * p0 + pi),
                       -1.0 / (1.0 - pi))
    g_direct[:, model.leaf['inflated_loc_probs'].offset] = np.sum(
        dlp_dpi * (pi * (1 - pi)), axis=-1)
    dl_dlogits = dl_dlogits * w
    dl_dtc = dl_dtc * w
  # logits = -log shape - log mean ; tc = 1/shape
  dmean = -dl_dlogits / mean
  dout = dmean * sigmoid(out)
  dshape = np.sum(-dl_dlogits / shape[:, None] - dl_dtc / (shape**2)[:, None],
                  axis=-1)
  g_direct[:, model.leaf['shape'].offset] = dshape * sigmoid(th_shape)
  return dout, g_direct


def map_loss_and_grad(model, theta, x, y, n_total, prior_weight=1.0,
                      dtype=np.float64, f32_trig_args=True, lik_scale=1.0):
  """loss (E,), grad (E,P) of  -(c*lik_scale*loglik + prior_weight*logprior).

  Backward follows SURVEY Appendix A.3 (hand-derived; checked against finite
  differences and torch autograd in tests/test_oracle_grad.py).
  """
  theta = np.asarray(theta, dtype=dtype)
  E = theta.shape[0]
  y = np.asarray(y, dtype=dtype)
  if y.ndim == 1:
    y = np.broadcast_to(y, (E,) + y.shape)
  out, ch = forward(model, theta, x, dtype, f32_trig_args, keep=True)
  B = y.shape[-1]
  c = dtype(n_total) / dtype(B) * dtype(lik_scale)
  W, L = model.width, model.depth
  ll = loglik(model, theta, out, y)
  loss = -(ll * c)
  g = np.zeros_like(theta)

  def put(name, val):
    lf = model.leaf[name]
    g[:, lf.offset:lf.offset + lf.size] += np.asarray(val).reshape(E, lf.size)

  dll_dout, g_direct = _dloglik_dout_and_params(model, theta, out, y)
  g += -c * g_direct
  dout = -c * dll_dout                                             # (E,B)

  # output layer
  os_ = model.view(theta, 'inv_sp_output_scale')
  put('inv_sp_output_scale', sigmoid(os_) * np.sum(dout * ch['v'], axis=-1))
  dv = ch['gam_o'][:, None] * dout
  HL = ch['Hs'][L]
  sW = np.sqrt(dtype(W))
  put(f'Dense_{L}/kernel', np.einsum('ebw,eb->ew', HL, dv) / sW)
  put(f'Dense_{L}/bias', np.sum(dv, axis=-1))
  ko = model.view(theta, f'Dense_{L}/kernel')[..., 0]
  dH = dv[:, :, None] * ko[:, None, :] / sW                        # (E,B,W)

  alpha = ch['alpha']
  dalpha = np.zeros(E, dtype=dtype)
  for l in range(L - 1, -1, -1):
    A = ch['As'][l]
    gam = ch['gammas'][l]
    th = np.tanh(A)
    el = _elu(A)
    dalpha += np.sum(dH * (el - th), axis=(1, 2))
    dact = alpha * np.where(A > 0, 1.0, np.exp(np.minimum(A, 0))) + (
        1 - alpha) * (1 - th * th)
    dA = dH * dact
    Z = A / gam
    ls = model.view(theta, f'inv_sp_layer_scale{l}')
    put(f'inv_sp_layer_scale{l}', sigmoid(ls) * np.sum(dA * Z, axis=(1, 2)))
    dZ = gam * dA
    Hl = ch['Hs'][l]
    n = Hl.shape[-1]
    sn = np.sqrt(dtype(n))
    put(f'Dense_{l}/kernel', np.matmul(np.swapaxes(Hl, 1, 2), dZ) / sn)
    put(f'Dense_{l}/bias', np.sum(dZ, axis=1))
    K = model.view(theta, f'Dense_{l}/kernel')
    dH = np.matmul(dZ, np.swapaxes(K, 1, 2)) / sn
  a1 = alpha[:, 0, 0]
  put('logit_activation_weight', a1 * (1 - a1) * dalpha)

  # features
  dH0 = dH
  u = ch['u']
  du = np.zeros_like(u)
  for gidx, (kind, arg, ncols, col0, sname) in enumerate(model.groups):
    Gg = ch['G'][gidx]
    dHg = dH0[..., col0:col0 + ncols]
    fs = model.view(theta, sname)
    put(sname, sigmoid(fs) * np.sum(dHg * Gg, axis=(1, 2)))
    dG = softplus(fs)[:, None, None] * dHg
    if kind == 'u':
      du += dG
    elif kind == 'fourier':
      deg = ncols // 2
      a = ch['fargs'][arg]
      ck = (2.0 * np.pi * 2.0**np.arange(deg)).astype(dtype)
      if f32_trig_args:
        ck = (np.float32(2.0 * np.pi) * np.float32(2.0)**np.arange(
            deg, dtype=np.float32)).astype(dtype)
      den = np.arange(1, deg + 1).astype(dtype)
      du[..., arg] += np.sum(
          ck * (-np.sin(a) * dG[..., :deg] + np.cos(a) * dG[..., deg:]) / den,
          axis=-1)
    elif kind == 'inter':
      for k, (p, q) in enumerate(model.interactions):
        du[..., p] += dG[..., k] * u[..., q]
        du[..., q] += dG[..., k] * u[..., p]
    # seasonal: data-constant, no gradient beyond its scale
  put('log_scale_adjustment', -np.sum(du * u, axis=1))

  if prior_weight != 0.0:
    pw = dtype(prior_weight)
    loss = loss - pw * log_prior(model, theta)
    z = theta - model.prior_loc().astype(dtype)
    g += pw * np.tanh(0.5 * z)
  return loss, g


# --------------------------------------------------------------------------
# Adam (optax.adam defaults)                      inference.py:580,605-606
# --------------------------------------------------------------------------
def adam_update(theta, m, v, g, t, lr, b1=0.9, b2=0.999, eps=1e-8):
  """t is the 1-based step count AFTER increment.  Returns new (theta, m, v)."""
  dt = theta.dtype.type
  m = dt(b1) * m + dt(1 - b1) * g
  v = dt(b2) * v + dt(1 - b2) * g * g
  mhat = m / dt(1 - b1**t)
  vhat = v / dt(1 - b2**t)
  theta = theta - dt(lr) * mhat / (np.sqrt(vhat) + dt(eps))
  return theta, m, v


# --------------------------------------------------------------------------
# initial values                          inference.py:399-427, 203-231
# --------------------------------------------------------------------------
def map_init(model, target, matrix_values, dtype=np.float64):
  """matrix_values (E, P): only the entries under matrix_mask() are used
  (TruncatedNormal(0,1,-2,2) draws supplied by the caller); everything else is
  0 except log_noise_scale = log(nanstd(y)/2)."""
  E = matrix_values.shape[0]
  theta = np.zeros((E, model.P), dtype=dtype)
  mm = model.matrix_mask()
  theta[:, mm] = matrix_values[:, mm]
  theta[:, model.leaf['log_noise_scale'].offset] = np.log(
      np.nanstd(np.asarray(target, dtype=np.float64)) / 2.0)
  return theta


def vi_init(model, matrix_values, dtype=np.float64):
  E = matrix_values.shape[0]
  mu = np.zeros((E, model.P), dtype=dtype)
  mm = model.matrix_mask()
  mu[:, mm] = matrix_values[:, mm]
  rho = np.full((E, model.P), np.log(np.expm1(0.3)), dtype=dtype)
  return mu, rho


# --------------------------------------------------------------------------
# training loops                                  inference.py:577-619
# --------------------------------------------------------------------------
def train_map(model, theta0, X, y, lr, num_epochs, batch_size=None,
              prior_weight=1.0, row_index_fn=None, dtype=np.float64,
              f32_trig_args=True, return_state=False):
  """Full restatement of ensemble_map's _run for E members.

  row_index_fn(epoch) -> int array (E, steps*batch) of shuffled row ids; it is
  only consulted when batch_size < N (inference.py:594-597).  The ragged tail
  is dropped (inference.py:583-589).  Returns (theta, losses (E, num_epochs)).
  """
  X = np.asarray(X, dtype=dtype)
  y = np.asarray(y, dtype=dtype)
  N = y.shape[0]
  B = N if batch_size is None else int(batch_size)
  steps = N // B
  theta = np.array(theta0, dtype=dtype)
  E = theta.shape[0]
  m = np.zeros_like(theta)
  v = np.zeros_like(theta)
  losses = np.zeros((E, num_epochs), dtype=dtype)
  t = 0
  for ep in range(num_epochs):
    if B < N:
      idx = np.asarray(row_index_fn(ep))
    acc = np.zeros(E, dtype=dtype)
    for s in range(steps):
      if B < N:
        rows = idx[:, s * B:(s + 1) * B]
        xb, yb = X[rows], y[rows]
      else:
        xb, yb = X, y
      loss, g = map_loss_and_grad(model, theta, xb, yb, N, prior_weight, dtype,
                                  f32_trig_args)
      t += 1
      theta, m, v = adam_update(theta, m, v, g, t, lr)
      acc += loss
    losses[:, ep] = acc / steps
  if return_state:
    return theta, losses, m, v
  return theta, losses


# --------------------------------------------------------------------------
# mean-field VI                                    inference.py:626-764
# --------------------------------------------------------------------------
def vi_sigma(rho):
  return 1e-4 + np.logaddexp(rho, 0.0)


def vi_loss_and_grad(model, mu, rho, eps, x, y, n_total, kl_weight,
                     dtype=np.float64, f32_trig_args=True):
  """eps (E,S,P) standard-normal draws.  Loss per member (UN-multiplied by
  kl_weight; ensemble_vi multiplies the trace afterwards, inference.py:758):

     mean_s [ log q(z_s) - log p(z_s) - (N/B) loglik(z_s) / kl_weight ]
  """
  mu = np.asarray(mu, dtype=dtype)
  rho = np.asarray(rho, dtype=dtype)
  eps = np.asarray(eps, dtype=dtype)
  E, S, P = eps.shape
  sig = vi_sigma(rho)
  loss = np.zeros(E, dtype=dtype)
  gmu = np.zeros_like(mu)
  grho_acc = np.zeros_like(mu)
  for s in range(S):
    z = mu + sig * eps[:, s]
    # -log p(z) - c*loglik(z)/kl_weight  ==  map loss with lik_scale=1/kl
    l_s, g_s = map_loss_and_grad(model, z, x, y, n_total, 1.0, dtype,
                                 f32_trig_args, lik_scale=1.0 / kl_weight)
    logq = np.sum(-0.5 * eps[:, s]**2 - np.log(sig) - 0.5 * LOG_2PI, axis=-1)
    loss += (logq + l_s) / S
    gmu += g_s / S
    grho_acc += g_s * eps[:, s] / S
  grho = sigmoid(rho) * (grho_acc - 1.0 / sig)
  return loss, gmu, grho


def train_vi(model, mu0, rho0, X, y, lr, num_steps, sample_size, kl_weight,
             eps_fn, batch_size=None, batch_index_fn=None, dtype=np.float64,
             f32_trig_args=True):
  """eps_fn(step) -> (E,S,P); batch_index_fn(step) -> (B,) row ids shared by
  every member (inference.py:704-709).  Returns mu, rho, losses (E,steps)
  already multiplied by kl_weight (inference.py:758)."""
  X = np.asarray(X, dtype=dtype)
  y = np.asarray(y, dtype=dtype)
  N = y.shape[0]
  mu = np.array(mu0, dtype=dtype)
  rho = np.array(rho0, dtype=dtype)
  mm, vm = np.zeros_like(mu), np.zeros_like(mu)
  mr, vr = np.zeros_like(mu), np.zeros_like(mu)
  E = mu.shape[0]
  losses = np.zeros((E, num_steps), dtype=dtype)
  for st in range(num_steps):
    if batch_size is not None and batch_size < N:
      rows = np.asarray(batch_index_fn(st))
      xb, yb = X[rows], y[rows]
    else:
      xb, yb = X, y
    loss, gmu, grho = vi_loss_and_grad(model, mu, rho, eps_fn(st), xb, yb, N,
                                       kl_weight, dtype, f32_trig_args)
    mu, mm, vm = adam_update(mu, mm, vm, gmu, st + 1, lr)
    rho, mr, vr = adam_update(rho, mr, vr, grho, st + 1, lr)
    losses[:, st] = loss * kl_weight
  return mu, rho, losses


# --------------------------------------------------------------------------
# predict + quantiles           inference.py:42-100, 103-200, 461-507
# --------------------------------------------------------------------------
def predict_normal(model, theta, X, dtype=np.float64, f32_trig_args=True,
                   batchsize=1024):
  """-> means (E, N*), scales (E,).  1024-row slices like inference.py:134."""
  X = np.asarray(X, dtype=dtype)
  outs = [forward(model, theta, X[i:i + batchsize], dtype, f32_trig_args)
          for i in range(0, X.shape[0], batchsize)]
  return np.concatenate(outs, axis=-1), noise_scale(
      model, np.asarray(theta, dtype=dtype))


def _ndtr(z):
  return 0.5 * _sp.erfc(-z / math.sqrt(2.0))


def chandrupatla(f, low, high, value_tol=1e-5, max_iter=60, pos_tol=1e-8):
  """Vectorised Chandrupatla root finder (tfp.math.find_root_chandrupatla
  semantics: returns the bracket end with the smaller |f|; each element stops
  when |f_best| <= value_tol, the bracket is below pos_tol, or max_iter)."""
  a = np.array(low, dtype=np.float64)
  b = np.array(high, dtype=np.float64)
  a, b = np.broadcast_arrays(a, b)
  a, b = a.copy(), b.copy()
  fa, fb = f(a), f(b)
  c, fc = a.copy(), fa.copy()
  t = np.full_like(a, 0.5)
  best = np.where(np.abs(fa) < np.abs(fb), a, b)
  fbest = np.where(np.abs(fa) < np.abs(fb), fa, fb)
  done = (np.abs(fbest) <= value_tol)
  for _ in range(max_iter):
    if np.all(done):
      break
    xn = a + t * (b - a)
    fn = f(xn)
    same = np.sign(fn) == np.sign(fa)
    c_new = np.where(same, a, b)
    fc_new = np.where(same, fa, fb)
    b_new = np.where(same, b, a)
    fb_new = np.where(same, fb, fa)
    a_new, fa_new = xn, fn
    a = np.where(done, a, a_new); fa = np.where(done, fa, fa_new)
    b = np.where(done, b, b_new); fb = np.where(done, fb, fb_new)
    c = np.where(done, c, c_new); fc = np.where(done, fc, fc_new)
    a_best = np.abs(fa) < np.abs(fb)
    best = np.where(done, best, np.where(a_best, a, b))
    fbest = np.where(done, fbest, np.where(a_best, fa, fb))
    with np.errstate(divide='ignore', invalid='ignore'):
      tol = pos_tol / np.abs(b - c)
      done = done | (tol > 0.5) | (fbest == 0) | (np.abs(fbest) <= value_tol)
      xi = (a - b) / (c - b)
      phi = (fa - fb) / (fc - fb)
      use_iqi = (phi * phi < xi) & ((1 - phi)**2 < 1 - xi)
      t_iqi = (fa / (fb - fa)) * (fc / (fb - fc)) + ((c - a) / (b - a)) * (
          fa / (fc - fa)) * (fb / (fc - fb))
    t = np.where(use_iqi, t_iqi, 0.5)
    t = np.minimum(np.maximum(t, tol), 1 - tol)
    t = np.where(np.isfinite(t), t, 0.5)
  return best


def count_forecast(model, theta, out):
  """NB / ZINB forecast parameters and moments from the network output `out`
  (E, N*) (inference.py:103-125, 271-295; TFP 0.24 NegativeBinomial / Mixture):

     total_count = 1 / softplus(theta_shape)
     logits      = -log softplus(theta_shape) - log softplus(out)
     NB:   mean = tc e^logits,  var = mean / sigmoid(-logits)
     ZINB: Mixture([1 - pi, pi], [NB, delta_0])
  -> dict(tc (E,1), logits, pi (E,1) or None, mean, stddev)."""
  theta = np.asarray(theta, dtype=np.float64)
  tc, logits = nb_logits_total_count(model, theta, np.asarray(out, dtype=np.float64))
  tc = tc[:, None]
  mean = tc * np.exp(logits)
  var = mean / sigmoid(-logits)
  pi = None
  if model.observation_model == 'ZINB':
    pi = sigmoid(model.view(theta, 'inflated_loc_probs'))[:, None]
    zmean = (1 - pi) * mean
    var = (1 - pi) * (var + mean * mean) - zmean * zmean
    mean = zmean
  return dict(tc=tc, logits=logits, pi=pi, mean=mean, stddev=np.sqrt(var))


def count_cdf(fc, x):
  """cdf of each member's (ZI)NB at x >= 0 (TFP: betainc(tc, 1 + x, sigmoid(-logits)),
  continuous in x; zero inflation adds the point mass at 0)."""
  F = _sp.betainc(np.broadcast_to(fc['tc'], fc['logits'].shape), 1.0 + x,
                  sigmoid(-fc['logits']))
  if fc['pi'] is not None:
    F = fc['pi'] + (1 - fc['pi']) * F
  return F


def count_quantile_via_root(fc, q):
  """inference.py:298-333: ceil of the Chandrupatla root of mean_e cdf_e(x) - q on
  [0, max mean + 1.1 rsqrt(1 - q) max stddev]; 0 where mean_e pmf_e(0) > q."""
  high = np.max(fc['mean']) + 1.1 / np.sqrt(1.0 - q) * np.max(fc['stddev'])
  n = fc['logits'].shape[-1]
  root = chandrupatla(lambda x: count_cdf(fc, x[None, :]).mean(axis=0) - q,
                      np.zeros(n), np.full(n, high))
  p0 = count_cdf(fc, np.zeros((1, n))).mean(axis=0)
  return np.ceil(np.where(p0 > q, 0.0, root))


def normal_quantile_via_root(means, scales, q):
  """means (..., N*), scales (...,) ; mixture over all leading axes
  (inference.py:42-52)."""
  means = np.asarray(means, dtype=np.float64)
  scales = np.asarray(scales, dtype=np.float64)
  mu = means.reshape(-1, means.shape[-1])
  sd = scales.reshape(-1, 1)
  lo = mu.min() - 5 * sd.max()
  hi = mu.max() + 5 * sd.max()
  n = mu.shape[-1]

  def f(x):
    return _ndtr((x[None, :] - mu) / sd).mean(axis=0) - q

  return chandrupatla(f, np.full(n, lo), np.full(n, hi))


def approximate_normal_quantile(means, scales, q):
  """inference.py:55-84."""
  means = np.asarray(means, dtype=np.float64)
  scales = np.asarray(scales, dtype=np.float64)
  mu = means.reshape(-1, means.shape[-1])
  sd = scales.reshape(-1, 1)
  mmean = mu.mean(axis=0)
  mscale = np.sqrt((sd**2 + mu**2).mean(axis=0) - mmean**2)
  return mmean + mscale * _sp.ndtri(q)


def mixture_cdf(means, scales, x):
  mu = np.asarray(means, dtype=np.float64).reshape(-1, np.shape(means)[-1])
  sd = np.asarray(scales, dtype=np.float64).reshape(-1, 1)
  return _ndtr((np.asarray(x)[None, :] - mu) / sd).mean(axis=0)
