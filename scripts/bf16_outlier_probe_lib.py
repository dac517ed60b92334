"""Why does one bf16 fit in twelve leave one of its eight members 5-17 % high in RMSE after 200 steps
(tests/test_gpu_fullsize.py::test_c2_bf16_predictive_rmse_within_2pct_of_fp32)?

The C2 problem of that test, trained in chunks of CHUNK steps with the parameters read back after
every chunk: one fp32 reference run, N_BF16 bf16 runs and N_FP32 further fp32 runs (which differ from
the reference only by the order of their f32 atomics).  For every run: per-member RMSE deviation from
the reference at the end; for every run with a member beyond 5 %: the loss trajectory of that member
against the reference's, the first chunk at which any parameter leaf of that member has left the band
the well-behaved runs stay in, and which leaf it is.

(shared pieces of scripts/bf16_outlier_probe.py and scripts/first_run_probe2.py)
"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from tests import test_gpu_fullsize as T     # noqa: E402
from bayesnf_amd.engine import Engine        # noqa: E402

STEPS, CHUNK, E = 200, 10, 8
X, y, scales = T._grid()
net = T._net(scales)
fwd = Engine(net, members=E, forward_only=True, row_capacity=4096, compute_dtype='fp32')
Xd = torch.tensor(X, dtype=torch.float32, device=fwd.device)


def rmse(th):
  loc, _ = fwd.forward(torch.tensor(th, dtype=torch.float32, device=fwd.device), Xd)
  torch.cuda.synchronize()
  return np.sqrt(np.mean((loc.cpu().numpy() - y[None, :]) ** 2, axis=1))


def fit(dtype, perturb=None, rng=None, **extra):
  """perturb: None | 'init' | 'chunk' -- fp32 runs whose parameters are multiplied by (1 + 2^-9 u), u ~ U(-1, 1)
  (the size of one bf16 rounding) once at the start / before every chunk of CHUNK steps: how far does the
  training dynamics itself spread perturbations of that size in 200 steps?"""
  eng = Engine(net, X=X, y=y, members=E, seed=13, learning_rate=0.005, compute_dtype=dtype.split('+')[0], **extra)
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  ckpt, losses = [], []
  for e0 in range(0, STEPS, CHUNK):
    if perturb == 'chunk' or (perturb == 'init' and e0 == 0):
      th = eng.get_params()
      eng.set_params(th * (1.0 + 2.0 ** -9 * rng.uniform(-1, 1, th.shape)).astype(np.float32))
    l = eng.train(e0, CHUNK)
    torch.cuda.synchronize()
    losses.append(l.cpu().numpy())
    ckpt.append(eng.get_params().copy())
  eng.close()
  return np.stack(ckpt), np.concatenate(losses, axis=1)      # (chunks, E, P), (E, STEPS)


def leaf_dev(ck, ref):
  """(chunks, E, n_leaves): max |delta| of every leaf relative to the leaf's max |value| in the reference."""
  out = np.zeros(ck.shape[:2] + (len(net.leaves),))
  for k, lf in enumerate(net.leaves):
    sl = slice(lf.offset, lf.offset + lf.size)
    scale = np.maximum(np.abs(ref[..., sl]).max(axis=-1), 1e-6)
    out[..., k] = np.abs(ck[..., sl] - ref[..., sl]).max(axis=-1) / scale
  return out


