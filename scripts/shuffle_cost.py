"""What the reference-stream minibatch shuffles cost per epoch: drawn on the device (bnf_row_keys) against drawn on
the host and uploaded (bnf_row_tables), at a C2-like and a C4-like size.  usage: python scripts/shuffle_cost.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from bayesnf_amd import jaxseed as J            # noqa: E402
from bayesnf_amd.engine import Engine           # noqa: E402
from tests import util                          # noqa: E402

for n_rows, members, batch, host_too in ((10232, 64, 1024, True), (2_000_000, 16, 65536, False)):
  net, model, X, y = util.make_problem(n_rows=n_rows, width=64, depth=1)
  eng = Engine(net, X=X, y=y, members=members, batch=batch, seed=0)
  eng.init_params(0.0)
  epochs = 4
  pk = J.map_permute_keys(0, 1, members, epochs)[0]
  t0 = time.perf_counter()
  sub = J.map_shuffle_subkeys(pk, n_rows)
  t_keys = time.perf_counter() - t0
  eng.set_row_keys(sub, epoch0=0)
  eng.debug_row_index(0, 0)                    # allocates the work buffers, draws epoch 0
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for ep in range(1, epochs):
    eng.debug_row_index(ep, 0)
  torch.cuda.synchronize()
  t_dev = (time.perf_counter() - t0) / (epochs - 1)
  line = f'{n_rows} rows x {members} members: sub keys of {epochs} epochs on the host {t_keys * 1e3:.1f} ms; device draw {t_dev * 1e3:.2f} ms per epoch'
  if host_too:
    t0 = time.perf_counter()
    J.map_row_tables(pk[:, :1], n_rows, batch)
    line += f'; host draw {(time.perf_counter() - t0) * 1e3:.1f} ms per epoch (+ upload of {members * (n_rows // batch) * batch * 4 / 1e6:.1f} MB)'
  print(line)
  eng.close()
