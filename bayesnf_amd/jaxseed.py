"""Initial parameters from the SAME random streams the reference draws them from.

`BayesianNeuralFieldMAP/MLE.fit(table, seed=jax.random.PRNGKey(k))` in the reference initialises
every Dense kernel with `tfd.TruncatedNormal(0, 1, -2, 2)` through
`tfd.JointDistributionCoroutine(...).sample(seed=key_e)` where `key_e` comes from
`jax.random.split(split(seed)[0], (devices, members))` (/root/reference/src/bayesnf/
inference.py:399-427, 432-441, 571-575).  This module computes those numbers on the host (numpy,
uint32 arithmetic), so that a fit with the same seed starts from the reference's own initial
parameters -- with full-batch training (no shuffles) the fit then runs on the reference's own numbers, and the
reference's golden predictions are reproduced through the GPU engine (tests/test_gpu_estimator.py; what the goldens
pin: 5 full-batch Adam steps for MAP / MLE, 2 optimisation steps for VI -- `vi_noise_keys`' step recurrence beyond
step 2 and the minibatch shuffle chain are extrapolations of the same source-read recurrences).  It is host glue executed once per fit (E x P numbers), not part
of the per-step hot path; the device generator (Philox, `bnf_init_params`) remains available with
`init_rng='philox'` / BNF_INIT_RNG=philox and is what bench.py uses.

Restated library behaviour (jax 0.4.26 `jax/_src/prng.py`, `jax/_src/random.py`; tfp 0.24
`internal/samplers.py`, `distributions/joint_distribution*.py`, `truncated_normal.py`):
  threefry2x32 (20 rounds, `jax_threefry_partitionable` off): counts iota(n) split in halves;
  split(key, n) = threefry(key, iota(2n)).reshape(n, 2);  fold_in(key, d) = threefry(key, [0, d]);
  uniform: f = bitcast(bits >> 9 | 0x3f800000) - 1, max(lo, f (hi - lo) + lo);
  truncated_normal: sqrt2 erfinv(uniform(erf(lo / sqrt2), erf(hi / sqrt2))) clipped to the open interval;
  JointDistribution sampling: key <- fold_in(key, sha512('JointDistribution') mod 2^32) once, then
  `sample_seed, key = split(key)` before every yielded distribution.
"""

from __future__ import annotations

import hashlib

import numpy as np
from scipy import special as _sp

_U32 = np.uint32
_ROTATIONS = ((13, 15, 26, 6), (17, 29, 16, 24))
_JD_SALT = int(hashlib.sha512(b'JointDistribution').hexdigest(), 16) & 0xFFFFFFFF


def as_key(seed) -> np.ndarray:
  """int k -> PRNGKey(k) = [k >> 32, k & 0xffffffff]; a length-2 array is taken as a key."""
  if isinstance(seed, (int, np.integer)):
    k = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.array([k >> 32, k & 0xFFFFFFFF], dtype=_U32)
  arr = np.asarray(seed).reshape(-1)
  if arr.size == 1:
    return as_key(int(arr[0]))
  if arr.size != 2:
    raise ValueError('seed must be an int or a length-2 uint32 array (jax.random.PRNGKey)')
  return arr.astype(np.uint64).astype(_U32)


def _threefry(key, x0, x1):
  with np.errstate(over='ignore'):
    ks = (_U32(key[0]), _U32(key[1]), _U32(_U32(key[0]) ^ _U32(key[1]) ^ _U32(0x1BD11BDA)))
    x0 = x0.astype(_U32) + ks[0]
    x1 = x1.astype(_U32) + ks[1]
    for group in range(5):
      for r in _ROTATIONS[group % 2]:
        x0 = x0 + x1
        x1 = ((x1 << _U32(r)) | (x1 >> _U32(32 - r))) ^ x0
      x0 = x0 + ks[(group + 1) % 3]
      x1 = x1 + ks[(group + 2) % 3] + _U32(group + 1)
  return x0, x1


def _bits(key, n: int) -> np.ndarray:
  """jax.random.bits(key, (n,), uint32) (original, non-partitionable layout)."""
  m = n + (n & 1)
  c = np.arange(m, dtype=_U32)
  c[n:] = 0
  y0, y1 = _threefry(key, c[:m // 2], c[m // 2:])
  return np.concatenate([y0, y1])[:n]


def split(key, n: int = 2) -> np.ndarray:
  return _bits(key, 2 * n).reshape(n, 2)


def fold_in(key, data: int) -> np.ndarray:
  y0, y1 = _threefry(key, np.zeros(1, _U32), np.array([data & 0xFFFFFFFF], dtype=_U32))
  return np.array([y0[0], y1[0]], dtype=_U32)


def truncated_normal_std(key, n: int) -> np.ndarray:
  """n draws of TruncatedNormal(0, 1, -2, 2) in float32, flat order of the leaf."""
  sqrt2 = np.float32(np.sqrt(2.0))
  a = np.float32(_sp.erf(np.float64(np.float32(-2.0) / sqrt2)))
  b = np.float32(_sp.erf(np.float64(np.float32(2.0) / sqrt2)))
  f = ((_bits(key, n) >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
  u = np.maximum(a, f * (b - a) + a)
  x = sqrt2 * _sp.erfinv(u.astype(np.float64)).astype(np.float32)
  lo = np.nextafter(np.float32(-2.0), np.float32(np.inf))
  hi = np.nextafter(np.float32(2.0), np.float32(-np.inf))
  return np.clip(x, lo, hi).astype(np.float32)


def member_keys(seed, world: int, per_device: int, split_index=None) -> np.ndarray:
  """(world, per_device, 2): `jax.random.split(init_seed, (devices, ensemble_size))` of
  ensemble_map, with fit_map's `fold_in(seed, i)` when num_splits > 1."""
  key = as_key(seed)
  if split_index is not None:
    key = fold_in(key, int(split_index))
  init_seed = split(key, 2)[0]
  return split(init_seed, world * per_device).reshape(world, per_device, 2)


def map_initial_params(net, keys: np.ndarray, log_noise_init: float) -> np.ndarray:
  """(len(keys), P) float32 packed initial parameters of MAP / MLE members (inference.py:399-427):
  log_noise_scale = log(nanstd / 2), Dense kernels ~ TN(0, 1, [-2, 2]) from the member's key,
  every other leaf 0.  `net.leaves` is the reference's leaf order (one split per leaf, also for
  the leaves whose initial value is deterministic)."""
  keys = np.asarray(keys, dtype=_U32).reshape(-1, 2)
  theta = np.zeros((keys.shape[0], net.P), dtype=np.float32)
  theta[:, net.by_name['log_noise_scale'].offset] = np.float32(log_noise_init)
  for e, key in enumerate(keys):
    key = fold_in(key, _JD_SALT)
    for lf in net.leaves:
      pair = split(key, 2)
      sample_seed, key = pair[0], pair[1]
      if len(lf.shape) == 2:
        theta[e, lf.offset:lf.offset + lf.size] = truncated_normal_std(sample_seed, lf.size)
  return theta


def map_leaf_keys(net, keys: np.ndarray) -> np.ndarray:
  """For `bnf_init_params_keys`: the per-leaf sample seeds of `map_initial_params`' chain, (len(keys), n_leaves, 2)
  uint32 -- all the host does; bits, erfinv and clipping happen on the device."""
  keys = np.asarray(keys, dtype=_U32).reshape(-1, 2)
  out = np.empty((keys.shape[0], len(net.leaves), 2), dtype=_U32)
  for e, key in enumerate(keys):
    key = fold_in(key, _JD_SALT)
    for i in range(len(net.leaves)):
      pair = split(key, 2)
      out[e, i], key = pair[0], pair[1]
  return out


_IID_SALT = int(__import__('hashlib').sha512(b'iid_sample_stateless').hexdigest(), 16) & 0xFFFFFFFF


def vi_initial_means(net, seed, world: int, per_device: int) -> np.ndarray:
  """(world, per_device, P) float32 initial surrogate means of ensemble_vi (inference.py:722-725 with
  make_vi_init :203-231): `init_seed = split(seed)[0]`; the (devices, E) sample of the initialiser is a
  vectorised JointDistribution sample -- keys = split(fold_in(init_seed, 'iid_sample_stateless'),
  devices * E) -- and every key runs the JointDistribution chain over two yields per leaf (the mean,
  then the deterministic inverse-softplus scale): TN(0, 1, [-2, 2]) for Dense kernels, 0 elsewhere.
  Determined against the reference's VI golden (oracle/jax_rng.py, tests/test_jax_rng.py)."""
  init_seed = split(as_key(seed), 2)[0]
  keys = split(fold_in(init_seed, _IID_SALT), world * per_device)
  mu = np.zeros((world * per_device, net.P), dtype=np.float32)
  for e, key in enumerate(keys):
    key = fold_in(key, _JD_SALT)
    for lf in net.leaves:
      pair = split(key, 2)
      mean_seed, key = pair[0], pair[1]
      key = split(key, 2)[1]            # the leaf's second yield (deterministic rho) consumes a split too
      if len(lf.shape) == 2:
        mu[e, lf.offset:lf.offset + lf.size] = truncated_normal_std(mean_seed, lf.size)
  return mu.reshape(world, per_device, net.P)


def vi_mean_leaf_keys(net, seed, world: int, per_device: int) -> np.ndarray:
  """For `bnf_init_params_keys` on a VI handle: the per-leaf mean seeds of `vi_initial_means`' chain,
  (world, per_device, n_leaves, 2) uint32."""
  init_seed = split(as_key(seed), 2)[0]
  keys = split(fold_in(init_seed, _IID_SALT), world * per_device)
  out = np.empty((world * per_device, len(net.leaves), 2), dtype=_U32)
  for e, key in enumerate(keys):
    key = fold_in(key, _JD_SALT)
    for i in range(len(net.leaves)):
      pair = split(key, 2)
      out[e, i], key = pair[0], pair[1]
      key = split(key, 2)[1]            # the leaf's second yield (deterministic rho) consumes a split too
  return out.reshape(world, per_device, len(net.leaves), 2)


# ----------------------------------------------------------------------------- VI noise keys
def _split_many(keys: np.ndarray, n: int) -> np.ndarray:
  """split(key, n) for an array of keys (..., 2) -> (..., n, 2)."""
  keys = np.asarray(keys, dtype=_U32)
  c = np.arange(2 * n, dtype=_U32)
  y0, y1 = _threefry((keys[..., 0, None], keys[..., 1, None]), c[:n], c[n:])
  return np.concatenate([y0, y1], axis=-1).reshape(keys.shape[:-1] + (n, 2))


def _fold_in_many(keys: np.ndarray, data: int) -> np.ndarray:
  keys = np.asarray(keys, dtype=_U32)
  y0, y1 = _threefry((keys[..., 0], keys[..., 1]), np.zeros((), _U32), np.asarray(data & 0xFFFFFFFF, dtype=_U32))
  return np.stack([y0, y1], axis=-1)


def _jd_leaf_keys(keys: np.ndarray, n_leaves: int) -> np.ndarray:
  """per-leaf sample seeds of one JointDistribution execution per key: (..., 2) -> (..., n_leaves, 2)."""
  k = _fold_in_many(keys, _JD_SALT)
  out = np.empty(k.shape[:-1] + (n_leaves, 2), dtype=_U32)
  for i in range(n_leaves):
    pair = _split_many(k, 2)
    out[..., i, :] = pair[..., 0, :]
    k = pair[..., 1, :]
  return out


_MINIMIZE_SALT = int(hashlib.sha512(b'minimize').hexdigest(), 16) & 0xFFFFFFFF


def vi_noise_keys(net, seed, world: int, rank: int, num_steps: int, sample_size: int):
  """Key tables of the reference's VI noise for device `rank` (ensemble_vi, inference.py:722-753, through
  tfp.vi.fit_surrogate_posterior_stateless / tfp.math.minimize_stateless; determined against the
  reference's VI golden, oracle/jax_rng.py):
     init_seed, opt_seed = split(seed); fit_seed, sample_seed = split(opt_seed)
     s = fold_in(split(fit_seed, devices)[rank], 'minimize'); before every step s = split(s)[0];
     the step's `sample_size` joint samples: keys = split(fold_in(s, 'iid_sample_stateless'), sample_size),
     each -> fold_in 'JointDistribution', one split per leaf
  -> uint32 (num_steps, sample_size, n_leaves, 2).  (Two steps are what the golden validates.)"""
  states = vi_step_seeds(seed, world, rank, num_steps)
  sample_keys = _split_many(_fold_in_many(states, _IID_SALT), sample_size)     # (steps, S, 2)
  return _jd_leaf_keys(sample_keys, len(net.leaves))


def vi_step_seeds(seed, world: int, rank: int, num_steps: int) -> np.ndarray:
  """(num_steps, 2): the seed of every optimisation step of device `rank` -- what tfp.math.minimize_stateless hands the
  variational loss: s_0 = fold_in(split(fit_seed, devices)[rank], 'minimize'), s_k = split(s_{k-1})[0].  Pinned (two
  steps) by the reference's VI golden through the reparameterisation noise derived from it (`vi_noise_keys`)."""
  opt_seed = split(as_key(seed), 2)[1]
  fit_seed = split(opt_seed, 2)[0]
  s = fold_in(split(fit_seed, world)[rank], _MINIMIZE_SALT)
  states = np.empty((num_steps, 2), dtype=_U32)
  for k in range(num_steps):
    s = split(s, 2)[0]
    states[k] = s
  return states


def vi_batch_subkeys(seed, world: int, rank: int, num_steps: int, n_rows: int) -> np.ndarray:
  """For `bnf_row_keys` on a VI handle: uint32 (num_steps, 1, rounds, 2), the sub keys of the sort rounds of the ONE
  minibatch permutation ensemble_vi draws per step and shares between the members of a device
  (/root/reference/src/bayesnf/inference.py:704-709: `jax.random.permutation(seed, arange(N))[:batch_size]` with the `seed`
  keyword tfp.vi.fit_surrogate_posterior_stateless passes `target_log_prob_fn`).
  ASSUMPTION, unpinned by any golden (the reference's VI golden is full batch): that keyword is the step's own seed
  s_k of `vi_step_seeds` -- the same value the surrogate's samples are drawn from (golden-pinned: the sampler receives
  s_k unsplit) -- tfp 0.24's `monte_carlo_variational_loss` source is not available in this environment.  The sort
  rounds themselves are jax's `_shuffle` as in `map_shuffle_subkeys`."""
  states = vi_step_seeds(seed, world, rank, num_steps)
  return map_shuffle_subkeys(states[None], n_rows)      # members axis = 1


def vi_batches(seed, world: int, rank: int, num_steps: int, n_rows: int, batch: int) -> np.ndarray:
  """int32 (num_steps, batch): the row ids of every step's shared batch (host restatement of `vi_batch_subkeys`)."""
  return permutations(vi_step_seeds(seed, world, rank, num_steps), n_rows)[:, :batch]


def vi_draw_keys(net, seed, world: int, rank: int, num_draws: int):
  """Key table of the posterior draws (inference.py:741-753): uint32 (num_draws, n_leaves, 2)."""
  opt_seed = split(as_key(seed), 2)[1]
  sample_seed = split(split(opt_seed, 2)[1], world)[rank]
  return _jd_leaf_keys(_split_many(fold_in(sample_seed, _IID_SALT), num_draws), len(net.leaves))


# ----------------------------------------------------------------------------- minibatch shuffles (MAP / MLE)
def _bits_many(keys: np.ndarray, n: int) -> np.ndarray:
  """jax.random.bits(key, (n,)) for an array of keys (M, 2) -> (M, n) uint32."""
  keys = np.asarray(keys, dtype=_U32).reshape(-1, 2)
  m = n + (n & 1)
  c = np.arange(m, dtype=_U32)
  c[n:] = 0
  y0, y1 = _threefry((keys[:, 0, None], keys[:, 1, None]), c[None, :m // 2], c[None, m // 2:])
  return np.concatenate([y0, y1], axis=1)[:, :n]


def permutations(keys: np.ndarray, n: int) -> np.ndarray:
  """`jax.random.permutation(key, jnp.arange(n))` for every key of (M, 2) -> int32 (M, n).
  jax/_src/random.py `_shuffle`: ceil(3 ln n / ln(2^32 - 1)) rounds of `key, sub = split(key)`, a STABLE
  sort of the current order by `bits(sub, (n,))`."""
  keys = np.asarray(keys, dtype=_U32).reshape(-1, 2)
  rounds = shuffle_rounds(n)
  x = np.broadcast_to(np.arange(n, dtype=np.int32), (keys.shape[0], n)).copy()
  for _ in range(rounds):
    pair = _split_many(keys, 2)
    keys, sub = pair[:, 0], pair[:, 1]
    order = np.argsort(_bits_many(sub, n), axis=1, kind='stable')
    x = np.take_along_axis(x, order, axis=1)
  return x


def map_permute_keys(seed, world: int, per_device: int, num_epochs: int, split_index=None) -> np.ndarray:
  """(world, per_device, num_epochs, 2): the `permute_seed` of every member and epoch of ensemble_map
  (/root/reference/src/bayesnf/inference.py:571-575, 593, 622; fit_map's `fold_in(seed, i)` :432-441):
  `opt_seed = split(seed)[1]`, member key = `split(opt_seed, (devices, members))[d, e]`, and per epoch
  `seed, permute_seed = split(seed)`.  A function of the GLOBAL member index only."""
  key = as_key(seed)
  if split_index is not None:
    key = fold_in(key, int(split_index))
  carry = split(split(key, 2)[1], world * per_device)
  out = np.empty((world * per_device, num_epochs, 2), dtype=_U32)
  for ep in range(num_epochs):
    pair = _split_many(carry, 2)
    carry, out[:, ep] = pair[:, 0], pair[:, 1]
  return out.reshape(world, per_device, num_epochs, 2)


def shuffle_rounds(n: int) -> int:
  """sort rounds of `jax.random.permutation` over n items (jax/_src/random.py `_shuffle`)."""
  return int(np.ceil(3 * np.log(max(1, n)) / np.log(np.iinfo(np.uint32).max)))


def map_shuffle_subkeys(permute_keys: np.ndarray, n_rows: int) -> np.ndarray:
  """For `bnf_row_keys`: permute_keys (members, n_epochs, 2) -> uint32 (n_epochs, members, rounds, 2), the `sub`
  of every `key, sub = split(key)` of the permutation's sort rounds -- all the host does per fit; the bits and
  the stable sorts happen on the device."""
  pk = np.asarray(permute_keys, dtype=_U32)
  members, n_epochs = pk.shape[0], pk.shape[1]
  rounds = shuffle_rounds(n_rows)
  out = np.empty((n_epochs, members, rounds, 2), dtype=_U32)
  carry = pk.reshape(-1, 2)
  for r in range(rounds):
    pair = _split_many(carry, 2)
    carry = pair[:, 0]
    out[:, :, r] = pair[:, 1].reshape(members, n_epochs, 2).transpose(1, 0, 2)
  return out


def map_row_tables(permute_keys: np.ndarray, n_rows: int, batch: int) -> np.ndarray:
  """Epoch shuffles for `bnf_row_tables`: permute_keys (members, n_epochs, 2) -> int32 (n_epochs, members,
  (n_rows // batch) * batch): the leading full batches of each member's permuted data set
  (`_reshape_to_batches` drops the ragged tail, inference.py:583-589)."""
  pk = np.asarray(permute_keys, dtype=_U32)
  members, n_epochs = pk.shape[0], pk.shape[1]
  keep = (n_rows // batch) * batch
  out = np.empty((n_epochs, members, keep), dtype=np.int32)
  for ep in range(n_epochs):
    out[ep] = permutations(pk[:, ep], n_rows)[:, :keep]
  return out


def leaf_offsets(net) -> np.ndarray:
  """int32 (n_leaves + 1): offsets of the packed leaves in the reference's order."""
  off = [lf.offset for lf in net.leaves] + [net.P]
  return np.asarray(off, dtype=np.int32)
