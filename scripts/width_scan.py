"""Step time / algorithmic TFLOP/s of a depth-2 MAP step at the widths of the reference's dataset
configs (256, 512, 768, 1024; scripts/dataset_config.py), C2 rows and features, bf16."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from bayesnf_amd.spec import NetSpec
from bayesnf_amd.engine import Engine

X, y, scales = bench.synthetic_grid()
for width, members in ((256, 64), (512, 64), (768, 32), (1024, 16), (500, 64), (700, 32)):
  kw = dict(bench.MODEL_KW); kw['width'] = width
  net = NetSpec(input_scales=scales, **kw)
  eng = Engine(net, X=X, y=y, members=members, compute_dtype='bf16', learning_rate=0.005, seed=1)
  eng.init_params(0.0)
  eng.train(0, 5); torch.cuda.synchronize()
  t0 = time.perf_counter(); eng.train(5, 30); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
  F = net.F
  flops = 6.0 * len(y) * (F * width + width * width + width) * members
  print(f'width {width:5d} members {members:3d}: {dt * 1e3:7.3f} ms/step  {members / dt:9.0f} member-steps/s  {flops / dt / 1e12:6.1f} TFLOP/s', flush=True)
  eng.close()
