"""SURVEY 8d's fp8 gate, measured: final loss and RMSE of the ensemble-mean prediction after N full-batch Adam steps from
identical initial parameters, 'fp8' with the fp8 W x W contractions (default) and without (BNF_FP8_CONTRACT=0: fp8 operand
storage only) and 'bf16', relative to the 'fp32' run -- C2's and C5's feature layouts and widths (the shapes of
tests/test_gpu_fp8.py::test_fp8_training_within_survey_8d_statistical_bars_of_fp32), several seeds and step counts."""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
from bayesnf_amd.engine import Engine     # noqa: E402
from oracle import bnf_oracle as O        # noqa: E402
from tests import util                    # noqa: E402

LAYOUTS = {
    'C2': dict(n_rows=4000, width=512, depth=2, periods=(4.0, 52.1775), harmonics=(2, 10), T=522),
    'C5': dict(n_rows=6000, width=256, depth=2, periods=(7.0, 30.4375, 365.25), harmonics=(3, 10, 10), T=2000, interactions=()),
    'C3': dict(n_rows=3500, width=512, depth=4, periods=(24.0, 168.0), harmonics=(4, 4), T=2160, interactions=()),     # (MAP on C3's network)
    'C4': dict(n_rows=2048, width=1024, depth=4, periods=(7.0, 365.25), harmonics=(3, 10), T=10000, interactions=()),
}
if len(sys.argv) > 3:
  LAYOUTS = {k: v for k, v in LAYOUTS.items() if k in sys.argv[3].split(',')}
steps_list = [int(s) for s in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['150', '600'])]
seeds = [int(s) for s in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['3', '4', '5'])]
for layout, kw in LAYOUTS.items():
  net, model, X, y = util.make_problem(**kw)
  for steps in steps_list:
    for seed in seeds:
      out, theta0 = {}, None
      for name, dt, env in (('fp32', 'fp32', '1'), ('bf16', 'bf16', '1'), ('fp8 copies', 'fp8', '0'), ('fp8 c8', 'fp8', '1')):
        os.environ['BNF_FP8_CONTRACT'] = env
        eng = Engine(net, X=X, y=y, members=8, seed=seed, learning_rate=0.005, compute_dtype=dt)
        eng.init_params(float(np.log(np.nanstd(y) / 2)))
        if theta0 is None:
          theta0 = eng.get_params()
        else:
          eng.set_params(theta0)
        losses = eng.train(0, steps).cpu().numpy()
        th = eng.get_params().astype(np.float64)
        pred = np.asarray(O.forward(model, th, X)).mean(axis=0)
        out[name] = (float(np.mean(losses[:, -1])), float(np.sqrt(np.mean((pred - y) ** 2))))
        eng.close()
      l32, r32 = out['fp32']
      print(f'{layout} steps {steps:4d} seed {seed}: fp32 loss {l32:.1f} rmse {r32:.4f} | ' + ' | '.join(
          f'{n}: loss {out[n][0] / l32 - 1:+.4f} rmse {out[n][1] / r32 - 1:+.4f}' for n in ('bf16', 'fp8 copies', 'fp8 c8')), flush=True)
