#!/usr/bin/env python
"""Step time in the launch-bound regime: BASELINE config C1 (W=256, depth 2, E=8) on the
100-row fixture size and on the C2 grid, eager launches from the C loop (bnf_train)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesnf_amd.engine import Engine
from bayesnf_amd.spec import NetSpec
import bench

X, y, scales = bench.synthetic_grid()
for rows in (100, 10232):
  for dt in ('bf16', 'fp32'):
    net = NetSpec(input_scales=scales, **dict(bench.MODEL_KW, width=256))
    eng = Engine(net, X=X[:rows], y=y[:rows], members=8, seed=0, compute_dtype=dt)
    eng.init_params(0.0)
    eng.train(0, 20); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.train(20, 500); torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
    print(json.dumps({'config': 'C1-like W=256 depth=2 E=8', 'rows': rows, 'dtype': dt, 'us_per_step': round(dt_s / 500 * 1e6, 1),
                      'graph': os.environ.get('BNF_GRAPH', '0')}), flush=True)
    eng.close()
