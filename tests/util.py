"""Shared builders for the GPU parity tests: a small synthetic spatiotemporal
problem, the product NetSpec and the oracle Model describing the same network."""
import numpy as np

from bayesnf_amd.spec import NetSpec
from oracle import bnf_oracle as O


def make_problem(n_rows=300, width=64, depth=2, seed=0, interactions=((0, 1), (1, 2)),
                 fourier_degrees=(5, 3, 2), periods=(4.0, 52.1775), harmonics=(2, 10),
                 T=104, observation_model='NORMAL'):
  rng = np.random.default_rng(seed)
  t = rng.integers(0, T, n_rows).astype(np.float64)
  t[0], t[1] = 0, T - 1
  lat, lon = rng.standard_normal(n_rows), rng.standard_normal(n_rows)
  X = np.stack([t, lat, lon], axis=1).astype(np.float32).astype(np.float64)
  y = (3 * np.sin(2 * np.pi * t / periods[0]) + np.sin(2 * np.pi * t / periods[-1]) +
       2 * lat * lon + 0.5 * rng.standard_normal(n_rows))
  if observation_model != 'NORMAL':   # counts with a fair share of zeros
    y = rng.poisson(np.exp(0.4 * y)) * (rng.random(n_rows) > 0.25)
  y = y.astype(np.float32).astype(np.float64)
  kw = dict(observation_model=observation_model, width=width, depth=depth, input_scales=[T - 1.0, 1.0, 1.0],
            fourier_degrees=list(fourier_degrees), interactions=[list(p) for p in interactions],
            seasonality_periods=list(periods), num_seasonal_harmonics=list(harmonics))
  return NetSpec(**kw), O.Model(**kw), X, y


# SURVEY 8d "Parity gates", fp32 class, verbatim -- held against BOTH f32-class engines ('fp32' = exact f32 MFMA, the
# default; 'fp32_split' = split-bf16 contractions): forward rel-err <= 1e-5, loss rel-err <= 1e-5, gradients rel-err <= 1e-4
# (vs the float64 oracle; per leaf, max |g - g_o| over the leaf's max |g_o|), parameters after 100 full-batch Adam steps
# rel-err <= 1e-3.  Measured (scripts/fp32_contract_diag.py, profiles/r06_fp32_contract_diag.txt): 'fp32' out <= 3.5e-7,
# loss <= 2.4e-7, worst leaf 8.9e-5 (log_scale_adjustment at W = 256: the Fourier argument 2 pi 2^k u amplifies the f32
# rounding of u), parameters <= 4.3e-6; 'fp32_split' 3.1e-6 / 2.4e-7 / 7.7e-5 / 2.7e-5.
FP32_GATE = dict(out=1e-5, loss=1e-5, grad=1e-4, params100=1e-3)
FP32_DTYPES = ('fp32', 'fp32_split')


def random_theta(model, E, seed=1, scale=0.5):
  """Generic (not init-like) parameters so every gradient path is exercised."""
  rng = np.random.default_rng(seed)
  th = scale * rng.standard_normal((E, model.P))
  th[:, model.leaf['log_noise_scale'].offset] = np.log(1.3) + 0.1 * rng.standard_normal(E)
  return th.astype(np.float32).astype(np.float64)


def rel_err(a, b):
  """max |a-b| / max |b| (scale-aware, robust to near-zero entries)."""
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def per_leaf_rel_err(model, g, g_ref):
  out = {}
  for lf in model.leaves:
    sl = slice(lf.offset, lf.offset + lf.size)
    ref = np.max(np.abs(g_ref[..., sl]))
    out[lf.name] = float(np.max(np.abs(g[..., sl] - g_ref[..., sl])) / max(ref, 1e-30)) \
        if ref > 0 else float(np.max(np.abs(g[..., sl])))
  return out
