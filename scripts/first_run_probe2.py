"""Order effects in scripts/bf16_outlier_probe.py: is it 'first fit of the process' or 'fp32 fits after bf16
fits'?  Sequence: fp32 x3, bf16 x3, fp32 x3 (200 steps in chunks of 10, as there); per-member RMSE of each, and
the same sequence once more with the parameters NOT read back between the chunks."""
import sys
import numpy as np
sys.path.insert(0, '.')
from scripts.bf16_outlier_probe_lib import fit, rmse, fwd

ref = None
for tag in ('fp32', 'fp32', 'fp32', 'bf16', 'bf16', 'bf16', 'fp32', 'fp32', 'fp32'):
  ck, loss = fit(tag)
  r = rmse(ck[-1])
  if ref is None:
    ref = r
  print(tag, 'RMSE / first', np.round(r / ref, 4).tolist(), 'final loss', np.round(loss[:3, -1], 1).tolist(), flush=True)
fwd.close()
