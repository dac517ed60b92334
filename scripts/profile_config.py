#!/usr/bin/env python
"""Per-kernel HIP-event table of one of scripts/bench_configs.py's configurations:
python scripts/profile_config.py 'C5/8 wind-like MAP (bf16)'"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from scripts.bench_configs import CONFIGS, grid   # noqa: E402
from bayesnf_amd.engine import Engine             # noqa: E402
from bayesnf_amd.spec import NetSpec              # noqa: E402

name = sys.argv[1]
gk, nk, ek, _ = CONFIGS[name]
X, y, scales = grid(**gk)
net = NetSpec(input_scales=scales, fourier_degrees=[5, 5, 5], interactions=[], **nk)
eng = Engine(net, X=X, y=y, seed=0, compute_dtype=__import__('os').environ.get('BNF_BENCH_DTYPE', 'bf16'), **ek)
eng.init_params(0.0 if ek['mode'] == 'vi' else float(np.log(np.nanstd(y) / 2)))
eng.train(0, 1)
torch.cuda.synchronize()
eng.profile('*')
eng.train(1, 3 if ek['mode'] != 'vi' else 12)
torch.cuda.synchronize()
prof = eng.profile_read()
tot = sum(v['avg_ms'] * v['calls'] for v in prof.values())
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['avg_ms'] * kv[1]['calls']):
  tf = v['flops'] / (v['avg_ms'] * 1e-3) / 1e12 if v['flops'] else 0.0
  print(f'{k:16s} avg {v["avg_ms"]*1e3:9.1f} us x{v["calls"]:4d}  {100*v["avg_ms"]*v["calls"]/tot:5.1f} %  {tf:7.1f} TFLOP/s')
eng.close()
