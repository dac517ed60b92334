#!/usr/bin/env python
"""Supplementary throughput of the per-GPU share of BASELINE.json's other configurations
(C3 VI, C4 minibatch MLE, C5 MAP) on ONE GPU, synthetic data of SURVEY.md 8(d).  Not the
headline metric (bench.py measures C2); prints one JSON line per configuration."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from bayesnf_amd.engine import Engine   # noqa: E402
from bayesnf_amd.spec import NetSpec    # noqa: E402


def grid(T, S, periods, keep=None, seed=1234):
  rng = np.random.default_rng(seed)
  lat, lon = rng.uniform(-1, 1, S), rng.uniform(-1, 1, S)
  lat, lon = (lat - lat.mean()) / lat.std(), (lon - lon.mean()) / lon.std()
  t = np.repeat(np.arange(T, dtype=np.float64), S)
  s = np.tile(np.arange(S), T)
  if keep is not None:
    t, s = t[:keep], s[:keep]
  X = np.stack([t, lat[s], lon[s]], axis=1)
  y = (3 * np.sin(2 * np.pi * t / periods[0]) + np.sin(2 * np.pi * t / periods[-1]) + 2 * lat[s] * lon[s] +
       0.5 * rng.standard_normal(t.size))
  return X, y, [T - 1.0, 1.0, 1.0]


CONFIGS = {
    # name: (grid args, net kwargs, engine kwargs, epochs)
    'C3/8 air_quality-like VI': (dict(T=2160, S=36, periods=[24, 168], keep=76192),
                                 dict(width=512, depth=4, seasonality_periods=[24, 168],
                                      num_seasonal_harmonics=[4, 4]),
                                 dict(mode='vi', members=16, batch=3500, vi_samples=5, kl_weight=0.2,
                                      learning_rate=0.01), 40),
    # the full 10^7-row grid, resident once per GPU (X 120 MB, seasonal table 1 GB)
    'C4/8 synthetic minibatch MLE': (dict(T=10000, S=1000, periods=[7, 365.25]),
                                   dict(width=1024, depth=4, seasonality_periods=[7, 365.25],
                                        num_seasonal_harmonics=[3, 10]),
                                   dict(mode='map', members=32, batch=65536, prior_weight=0.0,
                                        learning_rate=0.005), 0),
    'C5/8 wind-like MAP (bf16)': (dict(T=6574, S=12, periods=[7, 30.4375, 365.25], keep=71000),
                                  dict(width=256, depth=2, seasonality_periods=[7, 30.4375, 365.25],
                                       num_seasonal_harmonics=[3, 10, 10]),
                                  dict(mode='map', members=64, learning_rate=0.005), 20),
}


def main():
  only = sys.argv[1] if len(sys.argv) > 1 else None       # 'C3' | 'C4' | 'C5'
  for name, (gk, nk, ek, epochs) in CONFIGS.items():
    if only and not name.startswith(only):
      continue
    X, y, scales = grid(**gk)
    net = NetSpec(input_scales=scales, fourier_degrees=[5, 5, 5], interactions=[], **nk)
    eng = Engine(net, X=X, y=y, seed=0, compute_dtype=__import__('os').environ.get('BNF_BENCH_DTYPE', 'bf16'), **ek)
    eng.init_params(0.0 if ek['mode'] == 'vi' else float(np.log(np.nanstd(y) / 2)))
    B = ek.get('batch') or len(y)
    # (the VI engine, like tfp.vi.fit_surrogate_posterior, counts single steps, not passes)
    steps_per_epoch = 1 if ek['mode'] == 'vi' else len(y) // B
    if epochs == 0:
      # C4: an epoch is 152 steps; time 12 of them through the step entry point (same kernels;
      # debug_loss_and_grad runs forward + backward + the optimiser kernel with apply = 0)
      n_timed = 12
      eng.debug_loss_and_grad(0, 0)
      torch.cuda.synchronize()
      import ctypes as C
      from bayesnf_amd import _native
      grads = torch.empty((ek['members'], net.P), dtype=torch.float32, device=eng.device)
      loss1 = torch.empty((ek['members'],), dtype=torch.float32, device=eng.device)
      t0 = time.perf_counter()
      for s_ in range(n_timed):
        _native.check(eng.lib.bnf_debug_loss_and_grad(eng.handle, 0, s_, C.c_void_p(grads.data_ptr()),
                                                      C.c_void_p(loss1.data_ptr())), 'step')
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      steps = n_timed
      losses = loss1[:, None]
    else:
      eng.train(0, 1)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      losses = eng.train(1, epochs)
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      steps = epochs * steps_per_epoch
    S = ek.get('vi_samples', 1)
    flops = net.flops_per_member_step(B, S) * ek['members'] * steps
    print(json.dumps({'config': name, 'rows': len(y), 'F': net.F, 'members_on_this_gpu': ek['members'],
                      'batch': B, 'steps': steps, 'seconds': round(dt, 3),
                      'member_steps_per_s': round(ek['members'] * steps / dt, 1),
                      'algorithmic_tflops': round(flops / dt / 1e12, 1),
                      'final_loss_mean': float(losses[:, -1].mean().item())}), flush=True)
    eng.close()
    del eng
    torch.cuda.empty_cache()


if __name__ == '__main__':
  main()
