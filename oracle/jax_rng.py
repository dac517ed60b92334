"""Bit-level restatement of the random streams the reference's goldens are a function of.

THIS IS TEST INFRASTRUCTURE (see oracle/bnf_oracle.py).  The reference draws its initial
parameters through jax.random (threefry2x32) and TensorFlow Probability's seed plumbing:

  jax.random.PRNGKey / split / fold_in / bits / uniform / normal / permutation
                                          reference call sites: src/bayesnf/inference.py:38,400,
                                          434,571-575,593,622,706,722-725,747-753
  tfd.JointDistributionCoroutine.sample   inference.py:425-427 (MAP / MLE init), :203-231 (VI init)
  tfd.TruncatedNormal(0, 1, -2, 2).sample inference.py:416-423

Neither library is under /root/reference nor installable here (jax==0.4.26, jaxlib==0.4.26,
tensorflow-probability==0.24.0: requirements.Python3.10.14.txt:19,20,51), so their published
algorithms are restated:

  * Threefry-2x32, 20 rounds (Salmon et al., SC'11), as jax/_src/prng.py applies it:
    counts are split in two halves (first half -> word 0, second half -> word 1), an odd count is
    padded with one zero; `jax_threefry_partitionable` is False (the 0.4.26 default).
  * split(key, n)   = threefry(key, iota(2 n)) reshaped (n, 2)
  * fold_in(key, d) = threefry(key, [0, d])
  * bits(key, shape)= threefry(key, iota(size))
  * uniform(key, shape, lo, hi) = max(lo, f * (hi - lo) + lo), f = bitcast(bits >> 9 | 0x3f800000) - 1
  * normal          = sqrt(2) erfinv(uniform(nextafter(-1, 0), 1))
  * truncated_normal(lo, hi) = clip(sqrt(2) erfinv(uniform(erf(lo / sqrt 2), erf(hi / sqrt 2))), open interval)
  * permutation(key, n): one round (n < 2^32 / ...) of sort-by-random-bits: key, sub = split(key);
    stable sort of arange(n) by bits(sub, (n,))
  * TFP: sanitize_seed(seed, salt) = fold_in(seed, int(sha512(salt).hexdigest(), 16) & (2^32 - 1));
    JointDistributionCoroutine.sample(seed=key) (sample_shape (), use_vectorized_map=True) salts the
    key ONCE with the string 'JointDistribution', then before EVERY yielded distribution
    (Deterministic ones included) does `sample_seed, seed = split(seed)`;
    TruncatedNormal._sample_n draws a standard truncated normal on the standardised bounds with the
    backend's parameterised sampler (= jax.random.truncated_normal) in (flat batch, n) layout.

PIN STATUS.  jax / tfp are not importable here, so the chain above was first written from their
published behaviour and then DETERMINED against the reference's own golden files: of every
variant tried (salt strings, split order, iid_sample pre-steps; scripts/n1_chain_search.py keeps
the search) exactly this one reproduces tests/test_data/bnf-map.chickenpox.8.mini.pred.csv and
bnf-mle...pred.csv element-wise (max |yhat - golden| = 1.9e-6 / 5.0e-6 over the 100 training
rows; any other chain is off by ~0.1).  The VI chain (three streams: initial surrogate means,
optimisation noise, posterior draws -- bottom of this file) was determined the same way against
bnf-vi...pred.csv: 2.5e-6, the next best of ~1,300 candidates 7e-3 (scripts/n1_vi_chain_search.py).  Further known answers in tests/test_jax_rng.py:
split(PRNGKey(0)) and normal(PRNGKey(0), (10,)) / normal(PRNGKey(42)) as printed in the JAX
documentation.
"""

from __future__ import annotations

import hashlib

import numpy as np
from scipy import special as _sp

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
U32 = np.uint32


def _rotl(x, r):
  return (x << U32(r)) | (x >> U32(32 - r))


def threefry2x32(key, x0, x1):
  """key (2,) uint32; x0, x1 uint32 arrays of one shape -> (y0, y1)."""
  with np.errstate(over='ignore'):
    k0, k1 = U32(key[0]), U32(key[1])
    ks = (k0, k1, U32(k0 ^ k1 ^ U32(0x1BD11BDA)))
    x0 = np.asarray(x0, dtype=U32) + ks[0]
    x1 = np.asarray(x1, dtype=U32) + ks[1]
    for i in range(5):
      for r in _ROT[i % 2]:
        x0 = x0 + x1
        x1 = _rotl(x1, r)
        x1 = x0 ^ x1
      x0 = x0 + ks[(i + 1) % 3]
      x1 = x1 + ks[(i + 2) % 3] + U32(i + 1)
  return x0, x1


def threefry_2x32(key, count):
  """jax/_src/prng.py threefry_2x32: flat counts, halves -> the two words, odd sizes padded."""
  count = np.asarray(count, dtype=U32).ravel()
  odd = count.size % 2
  if odd:
    count = np.concatenate([count, np.zeros(1, U32)])
  h = count.size // 2
  y0, y1 = threefry2x32(key, count[:h], count[h:])
  out = np.concatenate([y0, y1])
  return out[:-1] if odd else out


def prng_key(seed: int):
  return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=U32)


def split(key, num=2):
  shape = (num,) if np.isscalar(num) else tuple(num)
  n = int(np.prod(shape))
  return threefry_2x32(key, np.arange(2 * n, dtype=U32)).reshape(shape + (2,))


def fold_in(key, data: int):
  return threefry_2x32(key, np.array([0, data & 0xFFFFFFFF], dtype=U32))


def random_bits(key, shape):
  n = int(np.prod(shape)) if len(shape) else 1
  return threefry_2x32(key, np.arange(n, dtype=U32)).reshape(shape)


def uniform(key, shape, minval=0.0, maxval=1.0):
  """float32, as jax.random.uniform."""
  bits = random_bits(key, shape)
  f = ((bits >> U32(9)) | U32(0x3F800000)).view(np.float32) - np.float32(1.0)
  lo, hi = np.float32(minval), np.float32(maxval)
  return np.maximum(lo, f * (hi - lo) + lo).astype(np.float32)


def normal(key, shape):
  lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
  u = uniform(key, shape, lo, 1.0)
  return (np.float32(np.sqrt(2)) * _sp.erfinv(u.astype(np.float64)).astype(np.float32)).astype(np.float32)


def truncated_normal(key, lower, upper, shape):
  sqrt2 = np.float32(np.sqrt(2))
  a = np.float32(_sp.erf(np.float64(np.float32(lower) / sqrt2)))
  b = np.float32(_sp.erf(np.float64(np.float32(upper) / sqrt2)))
  u = uniform(key, shape, a, b)
  out = sqrt2 * _sp.erfinv(u.astype(np.float64)).astype(np.float32)
  return np.clip(out, np.nextafter(np.float32(lower), np.float32(np.inf)),
                 np.nextafter(np.float32(upper), np.float32(-np.inf))).astype(np.float32)


def permutation(key, n: int):
  """jax.random.permutation(key, n) for n where one sort round suffices (n^3 < 2^32 - 1 ... the
  reference's formula: rounds = ceil(3 ln n / ln(2^32 - 1)))."""
  rounds = int(np.ceil(3 * np.log(max(1, n)) / np.log(np.iinfo(np.uint32).max)))
  x = np.arange(n)
  for _ in range(rounds):
    key, sub = split(key, 2)
    sort_keys = random_bits(sub, (n,))
    x = x[np.argsort(sort_keys, kind='stable')]
  return x


# --------------------------------------------------------------------------- TFP seed plumbing
def tfp_salt(salt: str) -> int:
  return int(hashlib.sha512(str(salt).encode("utf-8")).hexdigest(), 16) & 0xFFFFFFFF


def sanitize_seed(seed, salt=None):
  return fold_in(seed, tfp_salt(salt)) if salt is not None else seed


def jdc_sample_seeds(seed, n_dists: int, salt='JointDistribution'):
  """Seeds JointDistributionCoroutine.sample(seed=seed) hands to its n_dists yielded distributions."""
  seed = sanitize_seed(seed, salt)
  out = []
  for _ in range(n_dists):
    sample_seed, seed = split(seed, 2)
    out.append(sample_seed)
  return out


def tfd_truncated_normal_std(seed, shape):
  """tfd.TruncatedNormal(0, ones(shape), -2, 2).sample(seed=seed): (flat batch, n = 1) layout."""
  flat = int(np.prod(shape))
  return truncated_normal(seed, -2.0, 2.0, (flat, 1)).reshape(shape)


# --------------------------------------------------------------------------- the reference's init
def reference_member_keys(seed, n_members: int, split_index=None):
  """Per-member init keys of fit_map / ensemble_map (inference.py:432-441, 571-575):
  seed_i = fold_in(seed, i) when num_splits > 1, init_seed = split(seed_i)[0],
  keys = split(init_seed, (devices, members per device)) -- the flat order is the member order."""
  key = np.asarray(seed, dtype=U32)
  if split_index is not None:
    key = fold_in(key, int(split_index))
  init_seed = split(key, 2)[0]
  return split(init_seed, n_members)


def reference_map_init_matrices(model, seed, n_members: int, split_index=None):
  """(n_members, P) array holding the reference's TruncatedNormal initial Dense kernels at their
  packed offsets (zeros elsewhere): inference.py:399-427 through the TFP chain above.  `model` is
  an oracle Model (leaf list in the reference's order)."""
  keys = reference_member_keys(seed, n_members, split_index)
  mats = np.zeros((n_members, model.P), dtype=np.float64)
  for e in range(n_members):
    seeds = jdc_sample_seeds(keys[e], len(model.leaves))
    for i, lf in enumerate(model.leaves):
      if len(lf.shape) == 2:
        mats[e, lf.offset:lf.offset + lf.size] = tfd_truncated_normal_std(seeds[i], lf.shape).ravel()
  return mats


def reference_map_permutations(seed, n_members: int, n_epochs: int, n_rows: int, split_index=None):
  """The data-set permutation every member draws in every epoch of a minibatch MAP / MLE fit
  (ensemble_map, inference.py:571-575: `init_seed, opt_seed = split(seed)`, member seed =
  `split(opt_seed, (devices, members))[d, e]`; `_one_epoch` :593-597: `seed, permute_seed = split(seed)`,
  `permute_dataset` :35-39 = `jax.random.permutation(permute_seed, arange(N))`) -> int (n_members,
  n_epochs, n_rows).  The chain is read off the reference's source; no golden exercises it (the
  reference's goldens are full-batch), so it rests on the pinned `split` / `bits` restatements."""
  key = np.asarray(seed, dtype=U32)
  if split_index is not None:
    key = fold_in(key, int(split_index))
  opt_seed = split(key, 2)[1]
  seeds = split(opt_seed, n_members)
  out = np.empty((n_members, n_epochs, n_rows), dtype=np.int64)
  for e in range(n_members):
    s = seeds[e]
    for ep in range(n_epochs):
      s, permute_seed = split(s, 2)
      out[e, ep] = permutation(permute_seed, n_rows)
  return out


# --------------------------------------------------------------------------- the reference's VI chain
# ensemble_vi (inference.py:626-764), determined against tests/test_data/bnf-vi.chickenpox.8.mini.pred.csv
# the same way as the MAP chain (scripts/n1_vi_chain_search.py: of ~1,300 candidate chains exactly one
# reproduces the golden, to 2.5e-6; the next best is off by 7e-3):
#   init_seed, opt_seed   = split(seed)                                        :722
#   fit_seed, sample_seed = split(opt_seed)                                    :747
#   initial surrogate     : make_vi_init(prior)((devices, E), seed=init_seed)  :724 -- a vectorised
#       JointDistribution sample: keys = split(fold_in(init_seed, 'iid_sample_stateless'), devices * E), each
#       key runs the JointDistribution chain above over TWO yields per leaf (mean, then the Deterministic rho)
#   optimisation          : per device s = fold_in(split(fit_seed, devices)[d], 'minimize'); before every step
#       s = split(s)[0], and the step's S reparameterisation draws are a vectorised JointDistribution sample
#       with seed s: keys = split(fold_in(s, 'iid_sample_stateless'), S), each -> fold_in 'JointDistribution',
#       one split per leaf, jax.random.normal over the leaf's (E, ...) batch
#   posterior draws       : the same vectorised sample with seed split(sample_seed, devices)[d], n = num_samples
def _vec_jd_keys(seed, n):
  return split(fold_in(seed, tfp_salt('iid_sample_stateless')), n)


def _jd_leaf_normals(model, key, n_members):
  """One execution of the surrogate JointDistribution: (n_members, P) standard normals."""
  seeds = jdc_sample_seeds(key, len(model.leaves))
  eps = np.zeros((n_members, model.P), dtype=np.float32)
  for i, lf in enumerate(model.leaves):
    eps[:, lf.offset:lf.offset + lf.size] = normal(seeds[i], (n_members * lf.size,)).reshape(n_members, lf.size)
  return eps


def reference_vi_seeds(seed):
  """-> (init_seed, fit_seed of device 0, sample_seed of device 0) for a single-device run."""
  init_seed, opt_seed = split(np.asarray(seed, dtype=U32), 2)
  fit_seed, sample_seed = split(opt_seed, 2)
  return init_seed, split(fit_seed, 1)[0], split(sample_seed, 1)[0]


def reference_vi_init_means(model, seed, n_members: int):
  """(n_members, P) initial surrogate means: TruncatedNormal kernels, zeros elsewhere (inference.py:203-231)."""
  init_seed, _, _ = reference_vi_seeds(seed)
  keys = _vec_jd_keys(init_seed, n_members)
  mu = np.zeros((n_members, model.P), dtype=np.float64)
  for e in range(n_members):
    seeds = jdc_sample_seeds(keys[e], 2 * len(model.leaves))
    for i, lf in enumerate(model.leaves):
      if len(lf.shape) == 2:
        mu[e, lf.offset:lf.offset + lf.size] = tfd_truncated_normal_std(seeds[2 * i], lf.shape).ravel()
  return mu


def reference_vi_step_noise(model, seed, num_steps: int, sample_size: int, n_members: int):
  """List over steps of the (n_members, sample_size, P) reparameterisation noise of
  tfp.vi.fit_surrogate_posterior_stateless (two steps are what the golden validates)."""
  _, fit_seed, _ = reference_vi_seeds(seed)
  s = fold_in(fit_seed, tfp_salt('minimize'))
  out = []
  for _ in range(num_steps):
    s = split(s, 2)[0]
    keys = _vec_jd_keys(s, sample_size)
    out.append(np.stack([_jd_leaf_normals(model, keys[k], n_members) for k in range(sample_size)], axis=1))
  return out


def reference_vi_batches(seed, num_steps: int, n_rows: int, batch_size: int):
  """(num_steps, batch_size) row ids of ensemble_vi's per-step minibatch on a single device (inference.py:704-709:
  `jax.random.permutation(seed, arange(N))[:batch_size]`, one batch per step shared by the members).  `seed` is the
  keyword tfp.vi.fit_surrogate_posterior_stateless passes `target_log_prob_fn`: taken to be the step's own seed -- the
  value the golden-pinned reparameterisation draws of that step come from (`reference_vi_step_noise`).  UNPINNED by any
  golden (the VI golden is full batch) and an assumption about tfp 0.24's `monte_carlo_variational_loss`, whose source
  is not available here; the permutation itself rests on the pinned `split` / `random_bits` restatements."""
  _, fit_seed, _ = reference_vi_seeds(seed)
  s = fold_in(fit_seed, tfp_salt('minimize'))
  out = np.empty((num_steps, batch_size), dtype=np.int64)
  for k in range(num_steps):
    s = split(s, 2)[0]
    out[k] = permutation(s, n_rows)[:batch_size]
  return out


def reference_vi_posterior_noise(model, seed, num_samples: int, n_members: int):
  """(num_samples, n_members, P) standard normals of the posterior draws (inference.py:741-753)."""
  _, _, sample_seed = reference_vi_seeds(seed)
  keys = _vec_jd_keys(sample_seed, num_samples)
  return np.stack([_jd_leaf_normals(model, keys[k], n_members) for k in range(num_samples)])
