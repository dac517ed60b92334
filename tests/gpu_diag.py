"""Prints (never asserts) the stage-by-stage error table of the HIP engine vs
the oracle; run on the GPU box to see everything in one call:
    python tests/gpu_diag.py > gpurun_out/diag.txt"""
import json
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayesnf_amd.engine import Engine  # noqa: E402
from oracle import bnf_oracle as O  # noqa: E402
from tests import util  # noqa: E402


def section(name):
  print(f'\n=== {name} ===', flush=True)


def gemm_diag():
  net, model, X, y = util.make_problem(n_rows=8, width=64, depth=1)
  for dtype in ('fp32', 'bf16'):
    eng = Engine(net, X=X, y=y, members=1, compute_dtype=dtype)
    for (M, N, K) in [(128, 128, 64), (300, 200, 128), (57, 512, 320)]:
      rng = np.random.default_rng(M)
      A = rng.standard_normal((M, K)).astype(np.float32)
      Bt = rng.standard_normal((N, K)).astype(np.float32)
      Cd = eng.debug_gemm_nt(A, Bt)
      if dtype == 'bf16':
        A = torch.tensor(A).bfloat16().float().numpy()
        Bt = torch.tensor(Bt).bfloat16().float().numpy()
      ref = A.astype(np.float64) @ Bt.astype(np.float64).T
      err = util.rel_err(Cd, ref)
      # transposition / layout detectors
      err_t = util.rel_err(Cd, ref.T) if M == N else float('nan')
      print(f'gemm {dtype} {M}x{N}x{K}: rel_err {err:.3e}  (vs transposed {err_t:.3e})')
      if err > 1e-3:
        bad = np.argwhere(np.abs(Cd - ref) > 1e-3 * np.abs(ref).max())
        print('   first bad (m,n):', bad[:8].tolist(), ' n_bad', len(bad))
    eng.close()


def stage_diag(dtype, depth=2, width=64, n_rows=300, pw=1.0, pipeline='auto'):
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  E = 3
  theta = util.random_theta(model, E)
  eng = Engine(net, X=X, y=y, members=E, prior_weight=pw, compute_dtype=dtype, pipeline=pipeline)
  eng.set_params(theta)
  loss_d, g_d = eng.debug_loss_and_grad()
  out_o, ch = O.forward(model, theta, X, keep=True)
  loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows, prior_weight=pw)
  print(f'[{dtype} depth={depth} W={width} N={n_rows} pw={pw} pipeline={pipeline}]')
  print('  H0 max abs err', float(np.max(np.abs(eng.debug_activation(0) - ch['Hs'][0]))))
  for l in range(depth):
    try:
      print(f'  A{l} rel', util.rel_err(eng.debug_activation(100 + l), ch['As'][l]))
    except (RuntimeError, ValueError):
      print(f'  A{l} (kept on chip)')
    if l < depth - 1:
      print(f'  H{l+1} rel', util.rel_err(eng.debug_activation(1 + l), ch['Hs'][l + 1]))
  for l in range(depth):
    try:
      dz = eng.debug_activation(300 + l)
      print(f'  dZ{l} finite', bool(np.all(np.isfinite(dz))), 'max', float(np.abs(dz).max()))
    except (RuntimeError, ValueError):
      pass
  print('  out rel', util.rel_err(eng.debug_activation(200), out_o))
  print('  loss dev', loss_d, ' oracle', loss_o)
  errs = util.per_leaf_rel_err(model, g_d, g_o)
  for k, v in errs.items():
    print(f'  grad {k:28s} {v:.3e}')
  eng.close()


def train_diag(dtype):
  n_rows, E, steps = 200, 4, 30
  net, model, X, y = util.make_problem(n_rows=n_rows, width=64, depth=2)
  eng = Engine(net, X=X, y=y, members=E, seed=11, compute_dtype=dtype)
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  theta0 = eng.get_params().astype(np.float64)
  losses = eng.train(0, steps).cpu().numpy()
  theta_o, losses_o = O.train_map(model, theta0, X, y, lr=0.005, num_epochs=steps)
  print(f'[train {dtype}] loss rel err per step (max over members):',
        np.max(np.abs(losses - losses_o) / np.abs(losses_o), axis=0).round(7).tolist())
  print('  param rel err', util.rel_err(eng.get_params(), theta_o))
  print('  loss first/last', losses[0, 0], losses[0, -1], ' oracle', losses_o[0, 0], losses_o[0, -1])
  eng.close()


if __name__ == '__main__':
  print('torch', torch.__version__, 'cuda', torch.cuda.is_available(),
        torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)
  for name, fn in [('gemm', gemm_diag),
                   ('stages fp32', lambda: stage_diag('fp32')),
                   ('stages fp32 depth3 W192 N257', lambda: stage_diag('fp32', 3, 192, 257)),
                   ('stages fp32 mle', lambda: stage_diag('fp32', pw=0.0)),
                   ('stages bf16', lambda: stage_diag('bf16')),
                   ('train fp32', lambda: train_diag('fp32')),
                   ('train bf16', lambda: train_diag('bf16'))]:
    section(name)
    t0 = time.time()
    try:
      fn()
    except Exception:  # pylint: disable=broad-except
      traceback.print_exc(file=sys.stdout)
    print(f'({time.time() - t0:.1f}s)')
