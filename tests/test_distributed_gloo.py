"""N>1 path on CPU: two gloo ranks run a REAL sharded `fit_map` -- the product's own function, with the CPU stand-in
engine of tests/standin_engine.py injected through the one name `bayesnf_amd.inference.Engine` (compute itself is HIP-only;
the stand-in trains its members with the float64 oracle from the reference's initial particles) -- so that the members'
parameters and losses flow through run_shards -> gather_shards (one all-gather) -> _struct_tuple, and the gathered
ensemble equals the one-process fit of all members, full batch and minibatch (bayesnf_amd/distributed.py)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np
    import torch
    sys.path.insert(0, %r)
    from bayesnf_amd import distributed
    from bayesnf_amd.spec import NetSpec
    from bayesnf_amd import inference
    from bayesnf_amd import jaxseed as jaxseed_mod

    torch.distributed.init_process_group(backend='gloo')
    world, rank = distributed.device_count(), distributed.rank()
    assert world == 2 and distributed.is_distributed()
    E, R = 6, 17
    first, count = distributed.member_range(E)
    assert (first, count) == (rank * 3, 3)
    net = NetSpec(width=64, depth=1, input_scales=[9.0, 1.0], fourier_degrees=[2, 0],
                  interactions=[(0, 1)])
    # ---- a real sharded fit through the product's fit_map: stand-in engine behind the one seam ----
    from tests.standin_engine import StandInEngine
    kw = dict(width=64, depth=1, input_scales=[9.0, 1.0], fourier_degrees=[2, 0], interactions=[[0, 1]],
              seasonality_periods=[], num_seasonal_harmonics=[])
    StandInEngine.model_kwargs = dict(kw, observation_model='NORMAL')
    inference.Engine = StandInEngine
    rng = np.random.default_rng(0)
    X = np.stack([rng.integers(0, 10, 40).astype(float), rng.standard_normal(40)], axis=1)
    y = np.sin(X[:, 0]) + X[:, 1] + 0.1 * rng.standard_normal(40)
    args = dict(kw, init_x=X[:2])
    for bs in (None, 16):
      StandInEngine.created.clear()
      params, losses = inference.fit_map(X, y, 7, 'NORMAL', args, num_particles=E, learning_rate=0.01, num_epochs=3,
                                         batch_size=bs)
      assert StandInEngine.created == [(rank * 3, 3, 0)], StandInEngine.created      # this rank trained ITS members only
      assert losses.shape == (2, 3, 3) and params.var0.shape == (2, 3) and params[4].shape == (2, 3) + net.leaves[4].shape
      theta = inference._flatten_struct(net, params)                               # (2, 3, P), every rank holds all of it
      # the same fit in ONE shard (all six members on "device" 0 of a one-device job)
      eng = StandInEngine(net, X=X, y=y, batch=bs, members=E, learning_rate=0.01)
      eng.init_params_keys(jaxseed_mod.map_leaf_keys(net, jaxseed_mod.member_keys(7, 1, E)[0]), float(np.log(np.nanstd(y) / 2)))
      if bs is not None:
        eng.set_row_keys(jaxseed_mod.map_shuffle_subkeys(jaxseed_mod.map_permute_keys(7, 1, E, 3)[0], 40))
      l_one = eng.train(0, 3).numpy()
      np.testing.assert_array_equal(theta.reshape(E, net.P), eng.params.numpy().reshape(E, net.P))
      np.testing.assert_array_equal(losses.reshape(E, 3), l_one)
      assert np.abs(theta).max() > 0.5 and not np.array_equal(theta[0, 0], theta[1, 0])
    means_local = (np.arange(first, first + count)[:, None] * 10.0 + np.arange(R)[None, :]).astype(np.float32)
    means = distributed.all_gather_numpy(means_local)
    np.testing.assert_array_equal(means.reshape(E, R), np.arange(E)[:, None] * 10.0 + np.arange(R)[None, :])
    # real sharded work (host side of fit_map): every rank draws ITS members' initial parameters
    # from the reference's seed chain; the gathered ensemble equals the one-process result
    from bayesnf_amd import jaxseed
    keys = jaxseed.member_keys(7, world, 3)
    init_local = jaxseed.map_initial_params(net, keys[rank], 0.5)
    init_all = distributed.all_gather_stack(torch.from_numpy(init_local)).numpy()
    ref = jaxseed.map_initial_params(net, jaxseed.member_keys(7, 1, 6)[0], 0.5)
    np.testing.assert_array_equal(init_all.reshape(6, net.P), ref)
    assert np.abs(ref).max() > 0.5 and not np.array_equal(ref[0], ref[3])
    # ensemble_size smaller than the device count is rejected like the reference
    from bayesnf_amd import BayesianNeuralFieldMAP
    import pandas as pd
    df = pd.DataFrame({'t': pd.date_range('2020-01-06', periods=8, freq='W-MON'), 'y': range(8)})
    try:
      BayesianNeuralFieldMAP(feature_cols=['t'], target_col='y', freq='W', width=64).fit(
          df, seed=0, ensemble_size=1, num_epochs=1)
      raise SystemExit('expected ValueError')
    except ValueError:
      pass
    torch.distributed.barrier()
    if rank == 0:
      print('GLOO_OK')
    torch.distributed.destroy_process_group()
''') % ROOT


def test_two_rank_gather(tmp_path):
  script = tmp_path / 'worker.py'
  script.write_text(WORKER)
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', '29533', str(script)]
  out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
  assert 'GLOO_OK' in out.stdout


def test_member_range_floors_like_reference():
  from bayesnf_amd import distributed
  assert distributed.member_range(64, world=8, r=3) == (24, 8)
  assert distributed.member_range(10, world=4, r=3) == (6, 2)      # 10 // 4 = 2: two members dropped
  assert distributed.device_count() == 1 and distributed.rank() == 0
  a = np.arange(6, dtype=np.float32).reshape(2, 3)
  np.testing.assert_array_equal(distributed.all_gather_numpy(a), a[None])
