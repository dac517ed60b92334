"""Hunt for the rare wrong-gradient event of tests/test_gpu_parity.py::test_forward_and_grad_fp32.

Replays the test body (fresh engine -> set_params -> loss + gradient -> activation read-backs ->
close) for every parametrisation of the test, many times, against the oracle gradient computed once.
Any evaluation beyond the test's bar is dumped in detail: which leaves, which members, where inside
the leaf (row / column pattern), the ratio to the oracle, whether a second evaluation on the same
engine and a fresh engine reproduce it, and the env toggles in force.

usage: python scripts/flake_hunt2.py <seconds per case> [mode]
  mode: 'test' (default, the test body), 'noact' (no activation read-backs), 'one' (one engine, repeated)
"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from tests import util
from oracle import bnf_oracle as O
from bayesnf_amd.engine import Engine

CASES = [(2, 64, 300, 'layers'), (1, 128, 130, 'layers'), (3, 192, 257, 'layers'), (2, 256, 200, 'layers'),
         (2, 64, 300, 'auto'), (1, 128, 130, 'auto'), (3, 256, 257, 'auto'), (2, 192, 140, 'auto')]
BAR = 5e-4


def describe(model, g, g_ref, theta, pw):
  out = []
  for lf in model.leaves:
    sl = slice(lf.offset, lf.offset + lf.size)
    ref = np.max(np.abs(g_ref[..., sl]))
    d = np.abs(g[..., sl] - g_ref[..., sl]) / max(ref, 1e-30)
    if d.max() <= BAR:
      continue
    mem, idx = np.nonzero(d > BAR)
    shape = getattr(lf, 'shape', None)
    info = dict(leaf=lf.name, worst=float(d.max()), n_bad=int(len(idx)), size=int(lf.size), members=sorted(set(mem.tolist())),
                idx_min=int(idx.min()), idx_max=int(idx.max()), shape=str(shape))
    if shape is not None and len(shape) == 2:
      r, c = np.unravel_index(idx, shape)
      info.update(rows=[int(r.min()), int(r.max()), int(len(set(r.tolist())))], cols=[int(c.min()), int(c.max()), int(len(set(c.tolist())))])
    k = int(np.argmax(d.max(axis=0)))
    m = int(np.argmax(d[:, k]))
    gd, go = float(g[m, lf.offset + k]), float(g_ref[m, lf.offset + k])
    info.update(example=dict(member=m, index=k, got=gd, want=go, diff=gd - go, prior_term=float(pw * np.tanh(0.5 * theta[m, lf.offset + k]))))
    out.append(info)
  return out


def run_case(case, seconds, mode):
  depth, width, n_rows, pipeline = case
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  E = 3
  theta = util.random_theta(model, E)
  n_eval = n_bad = 0
  t_end = time.time() + seconds
  refs = {}
  for pw in (1.0, 0.0):
    loss_o, g_o = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows, prior_weight=pw)
    refs[pw] = (loss_o, g_o)
  eng = None
  while time.time() < t_end:
    for pw in (1.0, 0.0):
      loss_o, g_o = refs[pw]
      if mode != 'one' or eng is None:
        eng = Engine(net, X=X, y=y, members=E, prior_weight=pw if mode != 'one' else 1.0, compute_dtype='fp32', pipeline=pipeline)
        eng.set_params(theta)
      if mode == 'one':
        loss_o, g_o = refs[1.0]; pw = 1.0
      loss_d, g_d = eng.debug_loss_and_grad()
      n_eval += 1
      if mode == 'test':
        eng.debug_activation(0)
        for l in range(depth):
          if pipeline == 'layers' or (pipeline == 'auto' and 0 < l and (l < depth - 1 or width == 192)):
            eng.debug_activation(100 + l)
          if l < depth - 1:
            eng.debug_activation(1 + l)
        eng.debug_activation(200)
      errs = util.per_leaf_rel_err(model, g_d, g_o)
      lerr = float(np.max(np.abs(loss_d / loss_o - 1)))
      if max(errs.values()) > BAR or lerr > 2e-5:
        n_bad += 1
        loss_2, g_2 = eng.debug_loss_and_grad()
        eng2 = Engine(net, X=X, y=y, members=E, prior_weight=pw, compute_dtype='fp32', pipeline=pipeline)
        eng2.set_params(theta)
        loss_3, g_3 = eng2.debug_loss_and_grad()
        eng2.close()
        rec = dict(case=case, pw=pw, mode=mode, eval=n_eval, loss_rel=lerr, leaves=describe(model, g_d, g_o, theta, pw),
                   second_eval_bad=describe(model, g_2, g_o, theta, pw), second_eval_max_diff=float(np.abs(g_2 - g_d).max()),
                   fresh_engine_bad=describe(model, g_3, g_o, theta, pw))
        print('DEVIATION', json.dumps(rec), flush=True)
        np.savez(f'gpurun_out/flake_{depth}_{width}_{n_rows}_{pipeline}_{n_eval}.npz', g=g_d, g_ref=g_o, theta=theta, g2=g_2)
      if mode != 'one':
        eng.close()
  if eng is not None and mode == 'one':
    eng.close()
  print(f'case {case} mode {mode}: {n_eval} evaluations, {n_bad} deviating', flush=True)
  return n_eval, n_bad


if __name__ == '__main__':
  seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20
  mode = sys.argv[2] if len(sys.argv) > 2 else 'test'
  os.makedirs('gpurun_out', exist_ok=True)
  tot = bad = 0
  env = {k: v for k, v in os.environ.items() if k.startswith('BNF_')}
  print('env', env, flush=True)
  for case in CASES:
    a, b = run_case(case, seconds, mode)
    tot += a; bad += b
  print(f'TOTAL mode {mode}: {tot} evaluations, {bad} deviating', flush=True)
