"""Per-step wall time of the first 40 C2 steps of a fresh engine (synchronised after every step), twice (two engines)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
import bench                                    # noqa: E402
from bayesnf_amd.engine import Engine           # noqa: E402
from bayesnf_amd.spec import NetSpec            # noqa: E402

X, y, scales = bench.synthetic_grid()
net = NetSpec(input_scales=scales, **bench.MODEL_KW)
for rep in range(2):
  eng = Engine(net, mode='map', X=X, y=y, members=64, seed=0, learning_rate=0.005, prior_weight=1.0, compute_dtype='bf16')
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  out = []
  for ep in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train(ep, 1)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) * 1e3, 2))
  print('engine', rep, out)
  eng.close()
