#!/usr/bin/env python
"""C2 on ONE GPU as G concurrent engine handles of 64 / G members each (one host thread per handle, each handle its own
stream): does the hardware overlap one handle's HBM-bound weight-gradient / Adam kernels with another's issue-bound panel
kernel?  python scripts/two_handles_probe.py [G ...]   (profiles/r04_panel_ab.md r04y)"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench                                  # noqa: E402
from bayesnf_amd.engine import Engine         # noqa: E402
from bayesnf_amd.spec import NetSpec          # noqa: E402


def run(G, steps=40, warmup=8, total=64):
  X, y, scales = bench.synthetic_grid()
  net = NetSpec(input_scales=scales, **bench.MODEL_KW)
  E = total // G
  engs = []
  for g in range(G):
    eng = Engine(net, mode='map', X=X, y=y, members=E, member_offset=g * E, seed=0, learning_rate=0.005, prior_weight=1.0,
                 compute_dtype='bf16', device_index=0)
    eng.init_params(float(np.log(np.nanstd(y) / 2)))
    engs.append(eng)
  bar = threading.Barrier(G + 1)
  def work(eng):
    eng.train(0, warmup)
    torch.cuda.synchronize()
    bar.wait()
    eng.train(warmup, steps)
    torch.cuda.synchronize()
    bar.wait()
  th = [threading.Thread(target=work, args=(e,)) for e in engs]
  for t in th:
    t.start()
  bar.wait()
  t0 = time.perf_counter()
  bar.wait()
  dt = time.perf_counter() - t0
  for t in th:
    t.join()
  for e in engs:
    e.close()
  print(f'G={G} handles x {E} members: {dt / steps * 1e3:.4f} ms per step of {total} members, {total * steps / dt:.0f} member-steps/s', flush=True)


if __name__ == '__main__':
  for G in ([int(a) for a in sys.argv[1:]] or [1, 2, 4]):
    run(G)
