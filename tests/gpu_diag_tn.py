"""Mapping diagnostics of the transpose-read weight-gradient core (prints only)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesnf_amd.engine import Engine
from tests import util

net, model, X, y = util.make_problem(n_rows=8, width=64, depth=1)
eng = Engine(net, X=X, y=y, members=1, compute_dtype='bf16')
R, M, N = 64, 128, 128
I = np.zeros((R, N), np.float32); I[np.arange(R), np.arange(R)] = 1      # B[r][j] = (j == r)
Ar = np.repeat(np.arange(R, dtype=np.float32)[:, None], M, 1)            # A[r][i] = r
Ai = np.repeat(np.arange(M, dtype=np.float32)[None, :], R, 0)            # A[r][i] = i
for name, A in (('row', Ar), ('col', Ai)):
  C = eng.debug_gemm_tn(A, I)[:, :R]          # expect C[i][j] = A[j][i]
  exp = A.T[:, :R]
  bad = np.argwhere(C != exp)
  print(f'A-path {name}: mismatches {len(bad)} of {C.size}')
  for (i, j) in bad[:24]:
    print(f'   C[i={i}][j={j}] = {C[i, j]:.0f}  expected {exp[i, j]:.0f}')
# B path: A = identity on rows
IA = np.zeros((R, M), np.float32); IA[np.arange(R), np.arange(R)] = 1
Br = np.repeat(np.arange(R, dtype=np.float32)[:, None], N, 1)
Bj = np.repeat(np.arange(N, dtype=np.float32)[None, :], R, 0)
for name, B in (('row', Br), ('col', Bj)):
  C = eng.debug_gemm_tn(IA, B)[:R]            # expect C[i][j] = B[i][j]
  exp = B[:R]
  bad = np.argwhere(C != exp)
  print(f'B-path {name}: mismatches {len(bad)} of {C.size}')
  for (i, j) in bad[:24]:
    print(f'   C[i={i}][j={j}] = {C[i, j]:.0f}  expected {exp[i, j]:.0f}')
