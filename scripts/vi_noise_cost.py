"""Cost of the reference-compatible VI noise (threefry2x32 + erfinv on the device) against the engine's
own generator (Philox + Box-Muller): full-batch VI step, C2 rows and features, W = 512, S = 5."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from bayesnf_amd import jaxseed
from bayesnf_amd.spec import NetSpec
from bayesnf_amd.engine import Engine
X, y, scales = bench.synthetic_grid()
net = NetSpec(input_scales=scales, **bench.MODEL_KW)
for dt in ('bf16', 'fp32'):
  for mode in ('philox', 'jax'):
    eng = Engine(net, mode='vi', X=X, y=y, members=8, vi_samples=5, kl_weight=0.1, learning_rate=0.01, seed=1, compute_dtype=dt)
    eng.init_params(0.0)
    if mode == 'jax':
      eng.set_vi_noise_keys(jaxseed.vi_noise_keys(net, 0, 1, 0, 40, 5), None, jaxseed.leaf_offsets(net))
    eng.train(0, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.train(5, 30); torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 30
    print(f'{dt} {mode:7s} {dtm * 1e3:7.3f} ms/step', flush=True)
    eng.close()
