#!/usr/bin/env python
"""Search for the reference's VI seed chain (second part) against
tests/golden/bnf-vi.chickenpox.8.mini.pred.csv  (written by the reference's test_vi_mini:
seed PRNGKey(0), 1 particle, 2 steps, lr 0.01, kl_weight 0.1, 5 divergence samples, 30 posterior draws).

Stage 1 (this file, `stage1`): the golden's yhat is, to first order, a function of the INITIAL surrogate
means and of the 30 POSTERIOR draws only (two Adam steps of 0.01 move a mean by <= 0.02 against a
posterior scale of 0.3), so those two chains are searched jointly by correlation with yhat.
Stage 2 (`stage2`): with both fixed, the chain of the training noise is searched by max |yhat - golden|.
"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesnf_amd import spatiotemporal as st       # noqa: E402
from oracle import bnf_oracle as O                 # noqa: E402
from oracle import jax_rng as R                    # noqa: E402
from tests.test_oracle_kat import _load, _setup    # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')
model, X, y = _setup(G, st.BayesianNeuralFieldVI)
gold = _load(G, 'bnf-vi.chickenpox.8.mini.pred.csv').iloc[:100]
gy = gold.yhat.values
LEAVES = model.leaves
H = R.tfp_salt


def fold(k, salt):
  return k if salt is None else R.fold_in(k, H(salt))


def jd_seeds(seed, n, jd_salt='JointDistribution', order=0):
  """per-distribution seeds of one execution of a JointDistributionCoroutine"""
  s = fold(seed, jd_salt)
  out = []
  for _ in range(n):
    a, b = R.split(s, 2)
    ss, s = (a, b) if order == 0 else (b, a)
    out.append(ss)
  return out


BSPLIT = 0   # batch_ndims = 1 surrogates: 1 = every iid seed -> split(s, 1)[0]; 2 = split(seed, 1)[0] first


def vec_seeds(seed, n, outer, iid_salt, flat_split=True):
  """seeds of the n iid executions of a vectorised JD sample.
  outer: salt applied to the seed before the iid split (None / 'JointDistribution')"""
  if BSPLIT == 2:
    seed = R.split(seed, 1)[0]
  s = fold(seed, outer)
  s = fold(s, iid_salt)
  out = R.split(s, n)
  if BSPLIT == 1:
    out = np.stack([R.split(k, 1)[0] for k in out])
  return out


def init_means(seed, variant):
  outer, iid_salt, jd_salt, order, n_outer = variant
  key = R.split(seed, 2)[0]                       # init_seed, opt_seed = split(seed)
  if n_outer is None:
    mk = key
  else:
    mk = vec_seeds(key, 1, outer, iid_salt)[0]
  seeds = jd_seeds(mk, 2 * len(LEAVES), jd_salt, order)   # (mean, rho) per leaf
  mu = np.zeros(model.P)
  for i, lf in enumerate(LEAVES):
    if len(lf.shape) == 2:
      mu[lf.offset:lf.offset + lf.size] = R.tfd_truncated_normal_std(seeds[2 * i], lf.shape).ravel()
  return mu


def leaf_normals(seeds, per=1):
  eps = np.zeros(model.P)
  for i, lf in enumerate(LEAVES):
    eps[lf.offset:lf.offset + lf.size] = R.normal(seeds[per * i], (lf.size,))
  return eps


def posterior_draws(seed, variant, n=30):
  outer, iid_salt, jd_salt, order, dev_split = variant
  opt = R.split(seed, 2)[1]
  ss = R.split(opt, 2)[1]                         # fit_seed, sample_seed = split(opt_seed)
  if dev_split:
    ss = R.split(ss, 1)[0]                        # split(sample_seed, num_devices)[0]
  seeds = vec_seeds(ss, n, outer, iid_salt)
  return np.stack([leaf_normals(jd_seeds(seeds[k], len(LEAVES), jd_salt, order)) for k in range(n)])


def yhat_from(mu, rho, eps):
  th = mu[None] + O.vi_sigma(rho)[None] * eps
  return O.forward(model, th, X).mean(axis=0)


def stage1():
  seed = R.prng_key(0)
  rho0 = np.full(model.P, np.log(np.expm1(0.3)))
  iid_salts = [None, 'iid_sample_stateless', 'iid_sample']
  outers = [None, 'JointDistribution']
  jd_salts = ['JointDistribution', None]
  init_vars = [(o, i, j, od, 1) for o in outers for i in iid_salts for j in jd_salts for od in (0, 1)]
  init_vars += [(None, None, j, od, None) for j in jd_salts for od in (0, 1)]
  draw_vars = [(o, i, j, od, d) for o in outers for i in iid_salts for j in jd_salts for od in (0, 1) for d in (1, 0)]
  print(len(init_vars), 'init x', len(draw_vars), 'draw variants')
  mus = {v: init_means(seed, v) for v in init_vars}
  t0 = time.time()
  # the draws dominate yhat: rank the draw chains with ANY init first (correlation is driven by the draws
  # only through the common mu, so evaluate the full product but cheaply: 100 rows, 30 draws)
  res = []
  for dv in draw_vars:
    eps = posterior_draws(seed, dv)
    for iv in init_vars:
      yh = yhat_from(mus[iv], rho0, eps)
      r = np.corrcoef(yh, gy)[0, 1]
      res.append((r, np.abs(yh - gy).max(), iv, dv))
    print(dv, 'best so far', max(res)[:2], f'{time.time() - t0:.0f}s', flush=True)
  res.sort(key=lambda t: -t[0])
  for r in res[:12]:
    print(r)


INIT = (None, 'iid_sample_stateless', 'JointDistribution', 0, 1)   # stage 1: every one of the 12 best pairs has this init


def training_eps(seed, tv, steps=2, S=5):
  """tv = (dev_split, min_salt, step_order, mc_mode, mc_salt, outer, iid_salt, jd_salt)"""
  dev, ms, so, mcm, mcs, outer, iid, jd = tv
  opt = R.split(seed, 2)[1]
  fs = R.split(opt, 2)[0]
  if dev:
    fs = R.split(fs, 1)[0]
  s = fold(fs, ms)
  out = []
  for _ in range(steps):
    a, b = R.split(s, 2)
    step_seed, s = (a, b) if so == 0 else (b, a)
    if mcm == 'none':
      q = fold(step_seed, mcs)
    else:
      c, d = R.split(fold(step_seed, mcs), 2)
      q = c if mcm == 'first' else d
    seeds = vec_seeds(q, S, outer, iid)
    out.append(np.stack([leaf_normals(jd_seeds(seeds[k], len(LEAVES), jd, 0)) for k in range(S)])[None])
  return out


def stage2():
  seed = R.prng_key(0)
  mu0 = init_means(seed, INIT)[None]
  rho0 = np.full((1, model.P), np.log(np.expm1(0.3)))
  res = []
  t0 = time.time()
  vecs = [(o, i, j) for o in (None, 'JointDistribution') for i in ('iid_sample_stateless', None) for j in ('JointDistribution', None)]
  mcs = [('none', None)] + [(m, sl) for m in ('first', 'second') for sl in (None, 'monte_carlo_variational_loss')]
  tvs = [(dev, ms, so, mcm, mcsalt) + v for dev in (1, 0) for ms in (None, 'minimize_stateless', 'minimize')
         for so in (0, 1) for (mcm, mcsalt) in mcs for v in vecs]
  print(len(tvs), 'training variants')
  for n, tv in enumerate(tvs):
    eps = training_eps(seed, tv)
    mu, rho, _ = O.train_vi(model, mu0, rho0, X, y, lr=0.01, num_steps=2, sample_size=5, kl_weight=0.1,
                            eps_fn=lambda s_: eps[s_])
    for dsplit in (1, 0):
      dv = tv[5:7] + (tv[7], 0, dsplit)
      yh = yhat_from(mu[0], rho[0], posterior_draws(seed, dv))
      res.append((np.abs(yh - gy).max(), np.corrcoef(yh, gy)[0, 1], tv, dsplit))
    if n % 20 == 0:
      print(n, min(res)[:2], f'{time.time() - t0:.0f}s', flush=True)
  res.sort(key=lambda t: t[0])
  for r in res[:15]:
    print(r)


def stage3():
  """the best training chains of stage 2 x every posterior-draw chain (+ batch-dimension splits)"""
  global BSPLIT
  seed = R.prng_key(0)
  mu0 = init_means(seed, INIT)[None]
  rho0 = np.full((1, model.P), np.log(np.expm1(0.3)))
  V = (None, 'iid_sample_stateless', 'JointDistribution')
  tvs = [(1, 'minimize', 0, 'none', None) + V, (0, 'minimize', 1, 'second', None) + V,
         (0, 'minimize', 0, 'first', 'monte_carlo_variational_loss') + V, (1, None, 1, 'second', None) + V,
         (1, 'minimize_stateless', 0, 'second', 'monte_carlo_variational_loss') + V,
         (1, 'minimize', 0, 'first', None) + V, (1, 'minimize', 0, 'second', None) + V,
         (1, 'minimize_stateless', 0, 'none', None) + V, (1, None, 0, 'none', None) + V]
  draw_vars = [(o, i, j, od, d) for o in (None, 'JointDistribution') for i in (None, 'iid_sample_stateless', 'iid_sample')
               for j in ('JointDistribution', None) for od in (0, 1) for d in (1, 0)]
  res = []
  t0 = time.time()
  for tb in (0, 1, 2):
    for tv in tvs:
      BSPLIT = tb
      eps = training_eps(seed, tv)
      mu, rho, _ = O.train_vi(model, mu0, rho0, X, y, lr=0.01, num_steps=2, sample_size=5, kl_weight=0.1,
                              eps_fn=lambda s_: eps[s_])
      for db in (0, 1, 2):
        BSPLIT = db
        for dv in draw_vars:
          yh = yhat_from(mu[0], rho[0], posterior_draws(seed, dv))
          res.append((np.abs(yh - gy).max(), np.corrcoef(yh, gy)[0, 1], tb, tv[:5], db, dv))
      print(tb, tv[:5], min(res)[:2], f'{time.time() - t0:.0f}s', flush=True)
  res.sort(key=lambda t: t[0])
  for r in res[:15]:
    print(r)


def stage4():
  """init and posterior-draw chains fixed to the standard vectorised-JD chain; wider family of how the
  two optimisation steps get their seeds from fit_seed"""
  seed = R.prng_key(0)
  mu0 = init_means(seed, INIT)[None]
  rho0 = np.full((1, model.P), np.log(np.expm1(0.3)))
  V = (None, 'iid_sample_stateless', 'JointDistribution')
  eps_draw = {d: posterior_draws(seed, V + (0, d)) for d in (1, 0)}
  opt = R.split(seed, 2)[1]
  fit = R.split(opt, 2)[0]
  res = []
  t0 = time.time()
  def step_pairs(s):
    a, b = R.split(s, 2)
    t3 = R.split(s, 3)
    return {'a,split(b)0': (a, R.split(b, 2)[0]), 'a,split(b)1': (a, R.split(b, 2)[1]),
            'b,split(a)0': (b, R.split(a, 2)[0]), 'b,split(a)1': (b, R.split(a, 2)[1]),
            's,a': (s, a), 's,b': (s, b), 'a,b': (a, b), 'b,a': (b, a), 'a,a': (a, a), 's,s': (s, s),
            'fold0,fold1': (R.fold_in(s, 0), R.fold_in(s, 1)), 'fold1,fold2': (R.fold_in(s, 1), R.fold_in(s, 2)),
            'split3_01': (t3[0], t3[1]), 'split3_12': (t3[1], t3[2]),
            'a,split(a)0': (a, R.split(a, 2)[0]), 'b,split(b)0': (b, R.split(b, 2)[0]), 'b,split(b)1': (b, R.split(b, 2)[1])}
  def mc_variants(q):
    out = {'q': q, 'split0': R.split(q, 2)[0], 'split1': R.split(q, 2)[1]}
    f = fold(q, 'monte_carlo_variational_loss')
    out['salt'] = f; out['salt_split0'] = R.split(f, 2)[0]; out['salt_split1'] = R.split(f, 2)[1]
    return out
  def eps_for(q, S=5):
    seeds = vec_seeds(q, S, V[0], V[1])
    return np.stack([leaf_normals(jd_seeds(seeds[k], len(LEAVES), V[2], 0)) for k in range(S)])[None]
  n = 0
  for dev in (1, 0):
    base = R.split(fit, 1)[0] if dev else fit
    for ms in (None, 'minimize', 'minimize_stateless'):
      s0 = fold(base, ms)
      for pname, (q1, q2) in step_pairs(s0).items():
        m1, m2 = mc_variants(q1), mc_variants(q2)
        for mname in m1:
          eps = [eps_for(m1[mname]), eps_for(m2[mname])]
          mu, rho, _ = O.train_vi(model, mu0, rho0, X, y, lr=0.01, num_steps=2, sample_size=5, kl_weight=0.1,
                                  eps_fn=lambda s_: eps[s_])
          for d in (1, 0):
            yh = yhat_from(mu[0], rho[0], eps_draw[d])
            res.append((np.abs(yh - gy).max(), np.corrcoef(yh, gy)[0, 1], dev, ms, pname, mname, d))
          n += 1
          if n % 40 == 0:
            print(n, min(res)[:2], f'{time.time() - t0:.0f}s', flush=True)
  res.sort(key=lambda t: t[0])
  for r in res[:15]:
    print(r)


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == 'stage4':
    stage4()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'stage3':
    stage3()
    sys.exit(0)
  (stage2 if len(sys.argv) > 1 and sys.argv[1] == 'stage2' else stage1)()
