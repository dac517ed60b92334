import sys, os
sys.path.insert(0, '.')
import numpy as np
from tests import test_gpu_parity as T
fails = 0
cases = [(3, 192, 257, 'layers'), (2, 64, 300, 'layers'), (3, 256, 257, 'auto'), (2, 192, 140, 'auto')]
for it in range(int(sys.argv[1])):
  for c in cases:
    try:
      T.test_forward_and_grad_fp32(*c)
    except AssertionError as e:
      fails += 1
      print('FAIL iteration', it, c, str(e)[:600], flush=True)
print('iterations', sys.argv[1], 'x', len(cases), 'cases; failures', fails)
