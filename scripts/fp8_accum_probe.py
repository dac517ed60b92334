"""How exact is the accumulation of the non-scaled fp8 MFMA (v_mfma_f32_32x32x16_fp8_bf8) over a long K?  Operands that ARE
fp8 numbers (no conversion error), all positive (a truncating accumulator shows as a bias), through bnf_debug_gemm_tn of an
'fp8' handle (gemm_tn8) and of a 'bf16' handle (the same numbers are bf16 numbers too) -- against the float64 product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import util
from bayesnf_amd.engine import Engine

def q(a, kind):
  t = torch.tensor(np.asarray(a, dtype=np.float32))
  return (t.to(torch.float8_e4m3fn) if kind == 'e4m3' else t.to(torch.float8_e5m2)).float().numpy()

net, model, X, y = util.make_problem(n_rows=300, width=256, depth=2)
for dt in ('fp8', 'bf16'):
  eng = Engine(net, X=X, y=y, members=1, compute_dtype=dt)
  for R in (64, 1024, 16384, 131072):
    rng = np.random.default_rng(R)
    for name, lo, hi, sign in (('positive [0.5, 1)', 0.5, 1.0, False), ('signed, 4 decades', 1e-2, 1e2, True)):
      A = q(np.exp(rng.uniform(np.log(lo), np.log(hi), (R, 128))), 'e4m3')
      B = q(np.exp(rng.uniform(np.log(lo), np.log(hi), (R, 128))), 'e5m2')
      if sign:
        A *= rng.choice([-1.0, 1.0], A.shape); B *= rng.choice([-1.0, 1.0], B.shape)
      C = eng.debug_gemm_tn(A.astype(np.float32), B.astype(np.float32))
      ref = A.astype(np.float64).T @ B.astype(np.float64)
      d = (C - ref)
      print(f'{dt} R={R:6d} {name:20s}: max |err| / max |C| = {np.abs(d).max() / np.abs(ref).max():.2e}   mean signed err / mean |C| = {d.mean() / np.abs(ref).mean():+.2e}')
  eng.close()
