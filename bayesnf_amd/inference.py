"""Engine seam: same three entry points as the reference's inference module.

  fit_map      /root/reference/src/bayesnf/inference.py:376-458
  fit_vi       /root/reference/src/bayesnf/inference.py:336-373
  predict_bnf  /root/reference/src/bayesnf/inference.py:461-507

with identical argument names / meaning / return structure, so the estimator
layer (spatiotemporal.py) reads like the reference's.  Underneath, every member
lives on one GPU (`engine.Engine` -> libbnf_hip.so); ranks hold disjoint member
shards and the fitted parameters / predictive means are gathered once at the end (the
reference's implicit pmap output gather, inference.py:452,486-492).  Which devices a
process drives -- every visible GPU from one process like the reference's pmap, or one
per torchrun rank -- is `distributed.local_shards()`.
"""

from __future__ import annotations

import os
import warnings
from typing import Any

import numpy as np
import torch

from . import _native
from . import distributed
from . import jaxseed
from .engine import Engine
from .spec import NetSpec


def _net_from_args(model_args: dict[str, Any], observation_model: str) -> NetSpec:
  args = dict(model_args)
  args.pop('likelihood_distribution', None)
  return NetSpec(observation_model=observation_model, **args)


def _struct_tuple(net: NetSpec, theta: np.ndarray):
  """(..., P) -> StructTuple(var0, var1, ...) of (..., *leaf_shape) arrays."""
  return net.struct_tuple_type()(*net.unpack(theta))


def _flatten_struct(net: NetSpec, params) -> np.ndarray:
  """StructTuple with arbitrary leading dims -> (..., P) float32."""
  return net.pack(list(params), dtype=np.float32)


# ---------------------------------------------------------------------------
# MAP / MLE
# ---------------------------------------------------------------------------
def fit_map(features, target, seed, observation_model, model_args, num_particles,
            learning_rate, num_epochs, prior_weight=1.0, batch_size=None,
            num_splits=1, compute_dtype=None, init_rng=None):
  """Fit `num_particles` MAP (or MLE, prior_weight=0) members.

  init_rng: 'jax' (default; env BNF_INIT_RNG) starts every member from the initial parameters the
  reference itself would draw for `seed` (threefry + TFP seed chain restated on the host,
  `jaxseed`) and, for minibatch fits, shuffles every epoch with the reference's own per-member
  `jax.random.permutation` stream (`jaxseed.map_shuffle_subkeys` -> `bnf_row_keys`: drawn on the device): same seed => the same
  initial particles and shuffles as the reference (pinned by its goldens for full-batch fits; the shuffle chain rests on
  the reference's source + the pinned split / bits restatements -- no golden exercises a minibatch fit, and no end-to-end
  known-answer vector of `jax.random.permutation` is held: DESIGN.md section 5).  'philox' draws the initial parameters from the device generator
  (`bnf_init_params`) and shuffles with the device's keyed Feistel permutation (no index arrays).

  Returns (params, losses): params is a StructTuple whose leaves have shape
  (num_devices, num_particles // num_devices, *leaf_shape); losses has shape
  (num_devices, num_particles // num_devices, num_epochs).
  """
  net = _net_from_args(model_args, observation_model)
  features = np.asarray(features, dtype=np.float64)
  target = np.asarray(target, dtype=np.float64)
  n_rows = target.shape[0]
  if batch_size is None:
    batch_size = n_rows
  world = distributed.device_count()
  seed64 = _native.seed_to_u64(seed)
  per_device = (num_particles // num_splits) // world
  if per_device < 1:
    raise ValueError('fewer than one particle per device and split')
  log_noise_init = float(np.log(np.nanstd(target) / 2.0))
  init_rng = init_rng or os.environ.get('BNF_INIT_RNG', 'jax')
  if init_rng not in ('jax', 'philox'):
    raise ValueError("init_rng must be 'jax' or 'philox'")
  thetas, losses = [], []
  for i in range(num_splits):
    seed_i = _native.fold_in(seed64, i) if num_splits > 1 else seed64
    keys = (jaxseed.member_keys(seed, world, per_device, i if num_splits > 1 else None)
            if init_rng == 'jax' else None)

    # minibatch epochs under init_rng='jax': every member's per-epoch `jax.random.permutation` of the
    # reference (inference.py:593-597), keys on the host once per fit
    want_ref_shuffles = init_rng == 'jax' and batch_size < n_rows and num_epochs > 0
    if (want_ref_shuffles and per_device * n_rows >= 2**31
        and os.environ.get('BNF_ROW_TABLES', 'device') != 'host'):
      # the device-side sort addresses (member, row) pairs with 32-bit offsets (include/bnf.h bnf_row_keys)
      warnings.warn(f'{per_device} members x {n_rows} rows per device exceed 2^31 - 1: the epoch shuffles of this fit come '
                    "from the engine's index-free generator (same law as jax.random.permutation, other numbers); "
                    'BNF_ROW_TABLES=host keeps the reference stream at the price of host-drawn tables.')
      want_ref_shuffles = False
    pkeys = (jaxseed.map_permute_keys(seed, world, per_device, num_epochs, i if num_splits > 1 else None)
             if want_ref_shuffles else None)

    def train_shard(sh):
      # device sh.index of the job owns the members [index * per_device, (index + 1) * per_device): the
      # whole optimisation is enqueued on that device's stream; nothing is waited for here
      eng = Engine(net, mode='map', X=features, y=target, batch=batch_size,
                   members=per_device, member_offset=sh.index * per_device, seed=seed_i,
                   learning_rate=learning_rate, prior_weight=prior_weight,
                   compute_dtype=compute_dtype, device_index=sh.device)
      if keys is not None:
        # the reference's own initial particles for this seed: key chain on the host, values drawn on the device
        eng.init_params_keys(jaxseed.map_leaf_keys(net, keys[sh.index]), log_noise_init)
      else:
        eng.init_params(log_noise_init)
      if pkeys is None:
        return eng, eng.train(0, num_epochs)
      # the shuffles are drawn on the device, epoch by epoch, from the sub keys of their sort rounds
      # (BNF_ROW_TABLES=host: drawn here instead -- threefry bits + stable sort per member and epoch -- and uploaded
      # in chunks of <= ~64 MB of row ids, chunk c + 1 while the device runs the epochs of chunk c)
      if os.environ.get('BNF_ROW_TABLES', 'device') != 'host':
        eng.set_row_keys(jaxseed.map_shuffle_subkeys(pkeys[sh.index], n_rows), epoch0=0)
        return eng, eng.train(0, num_epochs)
      keep = (n_rows // batch_size) * batch_size
      per_chunk = int(max(1, min(num_epochs, (64 << 20) // max(1, 4 * per_device * keep))))
      parts = []
      for e0 in range(0, num_epochs, per_chunk):
        n = min(per_chunk, num_epochs - e0)
        eng.set_row_tables(jaxseed.map_row_tables(pkeys[sh.index][:, e0:e0 + n], n_rows, batch_size), epoch0=e0)
        parts.append(eng.train(e0, n))
      return eng, torch.cat(parts, dim=1)

    runs = distributed.run_shards(train_shard)
    thetas.append(distributed.gather_shards([e.params.view(per_device, net.P) for e, _ in runs]).cpu().numpy())
    losses.append(distributed.gather_shards([l for _, l in runs]).cpu().numpy())
    for e, _ in runs:
      e.close()
  theta = np.concatenate(thetas, axis=1)          # (world, E/world, P)
  return _struct_tuple(net, theta), np.concatenate(losses, axis=1)


# ---------------------------------------------------------------------------
# VI
# ---------------------------------------------------------------------------
class MeanFieldSurrogate:
  """What `ensemble_vi` returns first (a JointDistribution of Normals in the
  reference, inference.py:760-764): per-coordinate mean and scale."""

  def __init__(self, net: NetSpec, mu: np.ndarray, rho: np.ndarray):
    self.net = net
    self.loc = _struct_tuple(net, mu)
    self.scale = _struct_tuple(net, 1e-4 + np.logaddexp(rho, 0.0))
    self._mu, self._rho = mu, rho

  def mean(self):
    return self.loc

  def stddev(self):
    return self.scale


def fit_vi(features, target, seed, observation_model, model_args, ensemble_size,
           learning_rate, num_epochs, sample_size_divergence,
           sample_size_posterior, kl_weight, batch_size=None, compute_dtype=None, init_rng=None):
  """Fit mean-field surrogates.  Returns (surrogate, losses, predictions):
  losses (num_devices, E/num_devices, num_epochs) already multiplied by
  kl_weight; predictions = StructTuple of posterior draws with leaves
  (num_devices, sample_size_posterior, E/num_devices, *leaf_shape).

  init_rng='jax' (default): initial surrogate means, optimisation noise and posterior draws are the reference's own for
  `seed` (pinned by its VI golden: 2 full-batch steps; longer fits extrapolate the same key recurrence).  With `batch_size`
  the one row permutation every step shares between the device's members is the reference's too as far as its source shows
  it (inference.py:704-709) -- UNPINNED by any golden and resting on one stated assumption about tfp
  (jaxseed.vi_batch_subkeys); 'philox': the device generator throughout (same law, other numbers)."""
  net = _net_from_args(model_args, observation_model)
  init_rng = init_rng or os.environ.get('BNF_INIT_RNG', 'jax')
  if init_rng not in ('jax', 'philox'):
    raise ValueError("init_rng must be 'jax' or 'philox'")
  features = np.asarray(features, dtype=np.float64)
  target = np.asarray(target, dtype=np.float64)
  n_rows = target.shape[0]
  if batch_size is not None and n_rows < batch_size:
    raise AssertionError(f'batch_size={batch_size} exceeds target.shape[0]={n_rows}')
  world = distributed.device_count()
  per_device = ensemble_size // world
  if per_device < 1:
    raise ValueError('fewer than one surrogate per device')
  full_batch = batch_size is None or batch_size >= n_rows
  mu_keys = jaxseed.vi_mean_leaf_keys(net, seed, world, per_device) if init_rng == 'jax' else None

  def train_shard(sh):
    eng = Engine(net, mode='vi', X=features, y=target,
                 batch=n_rows if batch_size is None else batch_size,
                 members=per_device, member_offset=sh.index * per_device,
                 seed=_native.seed_to_u64(seed), learning_rate=learning_rate,
                 kl_weight=kl_weight, vi_samples=sample_size_divergence,
                 compute_dtype=compute_dtype, device_index=sh.device)
    if mu_keys is None:
      eng.init_params(0.0)
    else:
      # the reference's own initial surrogate means for this seed (key chain on the host, values on the device)
      eng.init_params_keys(mu_keys[sh.index], 0.0)
      # the optimisation noise and the posterior draws come from the reference's stream too (keys on the host once per
      # fit, normals on the device), so the fit runs on the reference's numbers (pinned by the reference's VI golden:
      # 2 full-batch optimisation steps; longer fits extrapolate the same key recurrence) ...
      eng.set_vi_noise_keys(jaxseed.vi_noise_keys(net, seed, world, sh.index, num_epochs, sample_size_divergence),
                            jaxseed.vi_draw_keys(net, seed, world, sh.index, sample_size_posterior),
                            jaxseed.leaf_offsets(net))
      if not full_batch:
        # ... and so does the ONE minibatch every step shares between the device's members: permutation(seed_step, N)[:B]
        # (inference.py:704-709), sort-round keys from the host, bits and sorts on the device.  Unpinned by any golden
        # (jaxseed.vi_batch_subkeys states the one assumption it rests on).
        eng.set_row_keys(jaxseed.vi_batch_subkeys(seed, world, sh.index, num_epochs, n_rows))
    loss_dev = eng.train(0, num_epochs)
    return eng, loss_dev, eng.vi_posterior_draws(sample_size_posterior)   # draws (n, E_local, P)

  runs = distributed.run_shards(train_shard)
  mu = distributed.gather_shards([e.params.view(2, per_device, net.P)[0] for e, _, _ in runs]).cpu().numpy()
  rho = distributed.gather_shards([e.params.view(2, per_device, net.P)[1] for e, _, _ in runs]).cpu().numpy()
  losses = distributed.gather_shards([l for _, l, _ in runs]).cpu().numpy()
  preds = distributed.gather_shards([d for _, _, d in runs]).cpu().numpy()   # (world, n, E/world, P)
  for e, _, _ in runs:
    e.close()
  return MeanFieldSurrogate(net, mu, rho), losses, _struct_tuple(net, preds)


# ---------------------------------------------------------------------------
# predict
# ---------------------------------------------------------------------------
_ROW_CHUNK = 8192


def _forward_local(net, theta_local, features, compute_dtype, device_index=None):
  """theta_local (M, P) numpy: members handled by one device -> loc (M, R), aux (M, 3)
  as device tensors, plus the engine (kept alive for the quantile kernels)."""
  n_rows = features.shape[0]
  M = theta_local.shape[0]
  # capacity: bound activation memory to ~2 GiB of (members x rows x width) cells
  cells = max(1, (1 << 28) // max(1, net.width))
  row_cap = int(min(n_rows, _ROW_CHUNK))
  mem_cap = int(max(1, min(M, cells // row_cap)))
  eng = Engine(net, mode='map', members=mem_cap, forward_only=True,
               row_capacity=row_cap, compute_dtype=compute_dtype, device_index=device_index)
  theta = torch.from_numpy(np.ascontiguousarray(theta_local, dtype=np.float32)).to(eng.device)
  X = torch.from_numpy(np.ascontiguousarray(
      np.asarray(features, dtype=np.float64), dtype=np.float32)).to(eng.device)
  loc, aux = eng.forward(theta, X)
  return eng, loc, aux


def _ensemble_forecast(features, observation_model, params, model_args,
                       ensemble_dims, compute_dtype):
  """Forward pass of every member on the new rows -> (net, engine, lead dims, loc (M, R), aux (M, 3))
  with all M members on the first local device.  The members are dealt out over the devices of the
  job in equal contiguous blocks whatever device count the parameters were fitted on (the leading
  dims of `params` only shape the result)."""
  net = _net_from_args(model_args, observation_model)
  theta_all = _flatten_struct(net, params)            # ([devices,] [S,] E/devices, P)
  lead = theta_all.shape[:-1]
  if len(lead) != ensemble_dims:
    raise ValueError(f'params have {len(lead)} ensemble dims, expected {ensemble_dims}')
  theta_flat = theta_all.reshape(-1, net.P)
  M = theta_flat.shape[0]
  world = distributed.device_count()
  per = -(-M // world)                                # ceil: the last block may be short

  def block(index):                                   # rows of theta_flat device `index` handles, padded to `per`
    lo = min(index * per, M - 1)
    idx = np.minimum(np.arange(lo, lo + per), M - 1)
    return theta_flat[idx]

  runs = distributed.run_shards(
      lambda sh: _forward_local(net, block(sh.index), features, compute_dtype, device_index=sh.device))
  loc_all = distributed.gather_shards([r[1] for r in runs])      # (world, per, R)
  aux_all = distributed.gather_shards([r[2] for r in runs])
  for r in runs[1:]:
    r[0].close()
  n_rows = features.shape[0]
  loc_all = loc_all.reshape(-1, n_rows)[:M]
  aux_all = aux_all.reshape(-1, 3)[:M]
  return net, runs[0][0], lead, loc_all, aux_all


def predict_bnf(features, observation_model, params, model_args, quantiles,
                ensemble_dims=2, approximate_quantiles=False, compute_dtype=None):
  """-> (means, [quantile arrays]).  means: leading ensemble dims of `params`
  + (n_rows,); each quantile array has shape (n_rows,)."""
  assert ensemble_dims >= 1
  features = np.asarray(features, dtype=np.float64)
  net, eng, lead, loc_all, aux_all = _ensemble_forecast(
      features, observation_model, params, model_args, ensemble_dims, compute_dtype)
  n_rows = features.shape[0]
  loc = loc_all.reshape(-1, n_rows)
  if observation_model == 'NORMAL':
    means = loc
    q = eng.normal_mixture_quantiles(means, aux_all.reshape(-1, 3)[:, 0], quantiles,
                                     approximate=approximate_quantiles)
  else:
    # NB / ZINB (inference.py:493-502): distribution means + root-found integer quantiles
    means, q = eng.count_mixture_quantiles(loc, aux_all.reshape(-1, 3), quantiles)
  torch.cuda.synchronize(eng.device)
  means_np = means.cpu().numpy().reshape(tuple(lead) + (n_rows,))
  q_np = q.cpu().numpy()
  eng.close()
  return means_np, [q_np[i] for i in range(q_np.shape[0])]


def _quantile_engine(net, obs, compute_dtype):
  """Forward-only handle that owns the quantile kernels (bnf_*_mixture_quantiles)."""
  return Engine(net, mode='map', members=1, forward_only=True, row_capacity=128, compute_dtype=compute_dtype)


class EnsembleLikelihood:
  """Stand-in for the TFP distribution returned by the reference's `likelihood_model`
  (spatiotemporal.py:433-468): Independent Normal per member with event shape (n_rows,) and
  batch shape = ensemble dims.  mean / stddev / log_prob / sample are per member (as TFP's);
  cdf is the per-member, per-row Normal cdf; mixture_cdf / quantile treat the ensemble as the
  equally weighted mixture `predict` reports quantiles of (inference.py:42-52), the root found
  on the GPU by the same kernel (`bnf_normal_mixture_quantiles`)."""

  def __init__(self, loc: np.ndarray, scale: np.ndarray, net=None, compute_dtype=None):
    self.loc = loc                       # (*ens, R)
    self.scale = scale[..., None]        # (*ens, 1)
    self._net, self._dtype = net, compute_dtype

  def mean(self):
    return self.loc

  def stddev(self):
    return np.broadcast_to(self.scale, self.loc.shape)

  def log_prob(self, y):
    y = np.asarray(y, dtype=np.float64)
    z = (y - self.loc) / self.scale
    return np.sum(-0.5 * z * z - np.log(self.scale) - 0.5 * np.log(2 * np.pi), axis=-1)

  def cdf(self, x):
    from scipy import special as sp
    return sp.ndtr((np.asarray(x, dtype=np.float64) - self.loc) / self.scale)

  def mixture_cdf(self, x):
    c = self.cdf(x)
    return c.reshape(-1, c.shape[-1]).mean(axis=0)

  def quantile(self, q, approximate=False):
    """Mixture quantile(s) per row: q scalar -> (R,), sequence -> (len(q), R)."""
    qs = np.atleast_1d(np.asarray(q, dtype=np.float64))
    eng = _quantile_engine(self._net, 'NORMAL', self._dtype)
    means = torch.from_numpy(np.ascontiguousarray(self.loc.reshape(-1, self.loc.shape[-1]), dtype=np.float32)).to(eng.device)
    scales = torch.from_numpy(np.ascontiguousarray(self.scale.reshape(-1), dtype=np.float32)).to(eng.device)
    out = eng.normal_mixture_quantiles(means, scales, qs.tolist(), approximate=approximate)
    torch.cuda.synchronize(eng.device)
    res = out.cpu().numpy().astype(np.float64)
    eng.close()
    return res[0] if np.ndim(q) == 0 else res

  def sample(self, seed=0):
    rng = np.random.default_rng(_native.seed_to_u64(seed))
    return self.loc + self.scale * rng.standard_normal(self.loc.shape)


class CountEnsembleLikelihood:
  """NB / ZINB counterpart (models.py:166-191): Independent (ZI)NegativeBinomial per member.
  total_count (*ens, 1), logits (*ens, R), inflated_loc_probs (*ens, 1) or None.  cdf /
  mixture_cdf / quantile as in EnsembleLikelihood; the integer quantile follows
  inference.py:298-333 (`bnf_count_mixture_quantiles`)."""

  def __init__(self, total_count, logits, inflated_loc_probs=None, net=None, compute_dtype=None, loc=None,
               aux=None):
    self.total_count = total_count[..., None]
    self.logits = logits
    self.inflated_loc_probs = None if inflated_loc_probs is None else inflated_loc_probs[..., None]
    self._net, self._dtype, self._loc, self._aux = net, compute_dtype, loc, aux

  def _nb_mean_var(self):
    mean = self.total_count * np.exp(self.logits)
    return mean, mean * (1.0 + np.exp(self.logits))       # mean / sigmoid(-logits)

  def mean(self):
    mean, _ = self._nb_mean_var()
    return mean if self.inflated_loc_probs is None else (1.0 - self.inflated_loc_probs) * mean

  def stddev(self):
    mean, var = self._nb_mean_var()
    if self.inflated_loc_probs is not None:
      pi = self.inflated_loc_probs
      var = (1.0 - pi) * (var + mean * mean) - ((1.0 - pi) * mean) ** 2
    return np.sqrt(var)

  def log_prob(self, y):
    from scipy import special as sp
    y = np.asarray(y, dtype=np.float64)
    tc, lg = self.total_count, self.logits
    lp = (tc * -np.logaddexp(lg, 0.0) + y * -np.logaddexp(-lg, 0.0) + sp.gammaln(tc + y) -
          sp.gammaln(1.0 + y) - sp.gammaln(tc))
    if self.inflated_loc_probs is not None:
      pi = self.inflated_loc_probs
      lp = np.where(y == 0, np.logaddexp(np.log1p(-pi) + lp, np.log(pi)), np.log1p(-pi) + lp)
    return np.sum(lp, axis=-1)

  def cdf(self, x):
    """P(Y <= x) per member and row (TFP: betainc(total_count, 1 + floor(x), sigmoid(-logits)))."""
    from scipy import special as sp
    x = np.floor(np.asarray(x, dtype=np.float64))
    tc = np.broadcast_to(self.total_count, self.logits.shape)
    c = np.where(x < 0, 0.0, sp.betainc(tc, 1.0 + np.maximum(x, 0.0), sp.expit(-self.logits)))
    if self.inflated_loc_probs is not None:
      c = np.where(x < 0, 0.0, self.inflated_loc_probs + (1.0 - self.inflated_loc_probs) * c)
    return c

  def mixture_cdf(self, x):
    c = self.cdf(x)
    return c.reshape(-1, c.shape[-1]).mean(axis=0)

  def quantile(self, q):
    qs = np.atleast_1d(np.asarray(q, dtype=np.float64))
    eng = Engine(self._net, mode='map', members=1, forward_only=True, row_capacity=128, compute_dtype=self._dtype)
    loc = torch.from_numpy(np.ascontiguousarray(self._loc.reshape(-1, self._loc.shape[-1]), dtype=np.float32)).to(eng.device)
    aux = torch.from_numpy(np.ascontiguousarray(self._aux.reshape(-1, 3), dtype=np.float32)).to(eng.device)
    _, out = eng.count_mixture_quantiles(loc, aux, qs.tolist())
    torch.cuda.synchronize(eng.device)
    res = out.cpu().numpy().astype(np.float64)
    eng.close()
    return res[0] if np.ndim(q) == 0 else res

  def sample(self, seed=0):
    rng = np.random.default_rng(_native.seed_to_u64(seed))
    shape = np.broadcast_shapes(self.total_count.shape, self.logits.shape)
    # NB(tc, p) as a Gamma-Poisson mixture: rate ~ Gamma(tc, scale = e^logits)
    rate = rng.gamma(np.broadcast_to(self.total_count, shape), np.exp(self.logits))
    draw = rng.poisson(rate).astype(np.float64)
    if self.inflated_loc_probs is not None:
      draw = np.where(rng.random(shape) < self.inflated_loc_probs, 0.0, draw)
    return draw


def likelihood_model(features, observation_model, params, model_args,
                     ensemble_dims=2, compute_dtype=None):
  features = np.asarray(features, dtype=np.float64)
  net, eng, lead, loc_all, aux_all = _ensemble_forecast(
      features, observation_model, params, model_args, ensemble_dims, compute_dtype)
  torch.cuda.synchronize(eng.device)
  n_rows = features.shape[0]
  loc = loc_all.cpu().numpy().reshape(tuple(lead) + (n_rows,)).astype(np.float64)
  aux = aux_all.cpu().numpy().reshape(tuple(lead) + (3,)).astype(np.float64)
  eng.close()
  if observation_model == 'NORMAL':
    return EnsembleLikelihood(loc, aux[..., 0], net=net, compute_dtype=compute_dtype)
  shape = aux[..., 1]
  logits = -np.log(shape)[..., None] - np.log(np.logaddexp(loc, 0.0))
  return CountEnsembleLikelihood(1.0 / shape, logits,
                                 aux[..., 2] if observation_model == 'ZINB' else None, net=net,
                                 compute_dtype=compute_dtype, loc=loc, aux=aux)
