"""Cold-start variant of scripts/flake_hunt2.py: ONE pass over the parametrisations of
tests/test_gpu_parity.py::test_forward_and_grad_fp32 per process (the pytest failure was a 1-in-~50
PROCESS event), LDS poisoned with NaN patterns before every evaluation, oracle gradients cached in
/tmp by the first process.  Run many of these, several at a time (they share the GPU, which also
perturbs the timing):  for i in $(seq 200); do python scripts/flake_hunt_cold.py & ...; done"""
import os, sys, json, pickle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from tests import util
from oracle import bnf_oracle as O
from bayesnf_amd.engine import Engine
from scripts.flake_hunt2 import CASES, BAR, describe

CACHE = '/tmp/flake_oracle_cache.pkl'
poison = int(os.environ.get('HUNT_POISON', '1'))
order = os.environ.get('HUNT_ORDER', 'test')
cases = list(CASES)
if order == 'shuffle':
  np.random.default_rng(os.getpid()).shuffle(cases)
refs = pickle.load(open(CACHE, 'rb')) if os.path.exists(CACHE) else {}
dirty = False
n_eval = n_bad = 0
for case in cases:
  depth, width, n_rows, pipeline = case
  net, model, X, y = util.make_problem(n_rows=n_rows, width=width, depth=depth)
  theta = util.random_theta(model, 3)
  for pw in (1.0, 0.0):
    key = (depth, width, n_rows, pw)
    if key not in refs:
      refs[key] = O.map_loss_and_grad(model, theta, X, y, n_total=n_rows, prior_weight=pw); dirty = True
    loss_o, g_o = refs[key]
    eng = Engine(net, X=X, y=y, members=3, prior_weight=pw, compute_dtype='fp32', pipeline=pipeline)
    eng.set_params(theta)
    if poison:
      eng.debug_poison_lds(0x7fc00000)
    loss_d, g_d = eng.debug_loss_and_grad()
    n_eval += 1
    errs = util.per_leaf_rel_err(model, g_d, g_o)
    lerr = float(np.max(np.abs(loss_d / loss_o - 1)))
    if not (max(errs.values()) <= BAR and lerr <= 2e-5):
      n_bad += 1
      loss_2, g_2 = eng.debug_loss_and_grad()
      print('DEVIATION', json.dumps(dict(case=case, pw=pw, pid=os.getpid(), eval=n_eval, loss_rel=lerr,
            leaves=describe(model, g_d, g_o, theta, pw), second_eval_bad=describe(model, g_2, g_o, theta, pw),
            second_eval_max_diff=float(np.abs(g_2 - g_d).max()))), flush=True)
      np.savez(f'gpurun_out/flake_cold_{os.getpid()}_{n_eval}.npz', g=g_d, g_ref=g_o, theta=theta, g2=g_2)
    eng.close()
if dirty:
  tmp = CACHE + f'.{os.getpid()}'
  pickle.dump(refs, open(tmp, 'wb')); os.replace(tmp, CACHE)
print(f'cold pid {os.getpid()}: {n_eval} evaluations, {n_bad} deviating', flush=True)
