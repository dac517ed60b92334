"""Does the C2 step get faster with the step COUNT (warm-up of something) or with TRAINING (data)?  Per-chunk step
times of one engine over 240 steps, then again after re-initialising the parameters of the same engine."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
import bench                                    # noqa: E402
from bayesnf_amd.engine import Engine           # noqa: E402
from bayesnf_amd.spec import NetSpec            # noqa: E402

X, y, scales = bench.synthetic_grid()
net = NetSpec(input_scales=scales, **bench.MODEL_KW)
eng = Engine(net, mode='map', X=X, y=y, members=64, seed=0, learning_rate=0.005, prior_weight=1.0, compute_dtype='bf16')
for label in ('fresh engine', 'same engine, parameters re-initialised', 'same engine, parameters kept (continues training)'):
  if not label.endswith('training)'):
    eng.init_params(float(np.log(np.nanstd(y) / 2)))
  out = []
  ep = 0
  for chunk in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train(ep, 20)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 20 * 1e3, 3))
    ep += 20
  print(label, out)
