"""Driver of the bf16 per-member outlier study; see scripts/bf16_outlier_probe_lib.py for the set-up.
usage: python scripts/bf16_outlier_probe.py [n_bf16] [n_fp32] > profiles/r03_bf16_outliers.txt"""
import json
import sys

import numpy as np

sys.path.insert(0, '.')
from scripts.bf16_outlier_probe_lib import *   # noqa: E402,F401,F403
from scripts.bf16_outlier_probe_lib import fit, rmse, leaf_dev, net, fwd, CHUNK

N_BF16 = int(sys.argv[1]) if len(sys.argv) > 1 else 96
N_FP32 = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ref_ck, ref_loss = fit('fp32')
ref_rmse = rmse(ref_ck[-1])
print('reference fp32 run: member RMSE', np.round(ref_rmse, 4).tolist(), 'final loss', np.round(ref_loss[:, -1], 1).tolist())
names = [lf.name for lf in net.leaves]
summary = {}
prng = np.random.default_rng(99)
for dtype, n in (('bf16', N_BF16), ('fp32', N_FP32), ('fp32+init', N_FP32), ('fp32+chunk', N_FP32)):
  devs, ldevs, runs = [], [], []
  for i in range(n):
    ck, loss = fit(dtype, perturb=(dtype.split('+')[1] if '+' in dtype else None), rng=prng)
    dev = rmse(ck[-1]) / ref_rmse - 1.0
    devs.append(dev)
    ld = leaf_dev(ck, ref_ck)
    ldevs.append(ld)
    runs.append((ck, loss, ld))
  devs = np.array(devs)                         # (n, E)
  ldevs = np.array(ldevs)                       # (n, chunks, E, leaves)
  calm = np.abs(devs).max(axis=1) < 0.03
  band = np.quantile(ldevs[calm].reshape(-1, ldevs.shape[1], len(names)), 0.99, axis=0) if calm.any() else None   # (chunks, leaves)
  n_out = int((np.abs(devs).max(axis=1) > 0.05).sum())
  summary[dtype] = dict(runs=n, runs_with_member_beyond_5pct=n_out, runs_with_member_beyond_8pct=int((np.abs(devs).max(axis=1) > 0.08).sum()),
                        member_dev_max=float(np.abs(devs).max()), member_dev_median_of_run_max=float(np.median(np.abs(devs).max(axis=1))),
                        per_member_beyond_5pct=[int(v) for v in (np.abs(devs) > 0.05).sum(axis=0)],
                        signed_mean_dev_per_member=[round(float(v), 4) for v in devs.mean(axis=0)])
  print(f'== {dtype}: {n} runs; runs with a member beyond 5 %: {n_out}; per member (count beyond 5 %): {summary[dtype]["per_member_beyond_5pct"]}')
  print('   run-max |dev| sorted tail:', np.round(np.sort(np.abs(devs).max(axis=1))[-8:], 4).tolist())
  for i in np.nonzero(np.abs(devs).max(axis=1) > 0.05)[0][:6]:
    m = int(np.abs(devs[i]).argmax())
    ck, loss, ld = runs[i]
    print(f'   outlier run {i}: member {m} RMSE dev {devs[i, m]:+.3f}; loss / reference loss of that member at steps 10, 20, ...:')
    print('     ', np.round(loss[m, CHUNK - 1::CHUNK] / ref_loss[m, CHUNK - 1::CHUNK], 4).tolist())
    step_max = int(np.argmax(loss[m] / ref_loss[m]))
    print(f'      largest loss ratio {float((loss[m] / ref_loss[m]).max()):.4f} at step {step_max}; '
          f'largest single-step loss increase of the member: {float(np.max(np.diff(loss[m]) / loss[m, :-1])):+.4f} '
          f'(reference member: {float(np.max(np.diff(ref_loss[m]) / ref_loss[m, :-1])):+.4f})')
    if band is not None:
      over = ld[:, m, :] > 3.0 * np.maximum(band, 1e-6)      # (chunks, leaves)
      first = [(int(np.argmax(over[:, k])) if over[:, k].any() else None) for k in range(len(names))]
      order = sorted([(c, names[k], float(ld[c, m, k]), float(band[c, k])) for k, c in enumerate(first) if c is not None])[:6]
      print('      first leaves to leave 3 x the 99 % band of the calm runs (chunk, leaf, deviation, band):')
      for c, nm, d, b in order:
        print(f'        after step {(c + 1) * CHUNK:3d}: {nm:28s} {d:.4f} (band {b:.4f})')
print(json.dumps(summary))
fwd.close()
