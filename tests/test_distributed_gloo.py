"""N>1 path on CPU: two gloo ranks exercise the member sharding arithmetic and the
posterior gather used by fit_map / predict_bnf (bayesnf_amd/distributed.py).  The
compute itself is HIP-only, so each rank fabricates its shard's "fitted"
parameters / predictive means as a deterministic function of the GLOBAL member id;
the gathered result must equal the single-process result."""
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

    torch.distributed.init_process_group(backend='gloo')
    world, rank = distributed.device_count(), distributed.rank()
    assert world == 2 and distributed.is_distributed()
    E, R = 6, 17
    first, count = distributed.member_range(E)
    assert (first, count) == (rank * 3, 3)
    net = NetSpec(width=64, depth=1, input_scales=[9.0, 1.0], fourier_degrees=[2, 0],
                  interactions=[(0, 1)])
    gid = np.arange(first, first + count)
    theta_local = (gid[:, None] * 1000 + np.arange(net.P)[None, :]).astype(np.float32)
    means_local = (gid[:, None] * 10.0 + np.arange(R)[None, :]).astype(np.float32)
    theta = distributed.all_gather_stack(torch.from_numpy(theta_local)).numpy()
    means = distributed.all_gather_numpy(means_local)
    assert theta.shape == (2, 3, net.P) and means.shape == (2, 3, R)
    full = (np.arange(E)[:, None] * 1000 + np.arange(net.P)[None, :]).astype(np.float32)
    np.testing.assert_array_equal(theta.reshape(E, net.P), full)
    np.testing.assert_array_equal(means.reshape(E, R),
                                  np.arange(E)[:, None] * 10.0 + np.arange(R)[None, :])
    # real sharded work (host side of fit_map): every rank draws ITS members' initial parameters
    # from the reference's seed chain; the gathered ensemble equals the one-process result
    from bayesnf_amd import jaxseed
    keys = jaxseed.member_keys(7, world, 3)
    init_local = jaxseed.map_initial_params(net, keys[rank], 0.5)
    init_all = distributed.all_gather_stack(torch.from_numpy(init_local)).numpy()
    ref = jaxseed.map_initial_params(net, jaxseed.member_keys(7, 1, 6)[0], 0.5)
    np.testing.assert_array_equal(init_all.reshape(6, net.P), ref)
    assert np.abs(ref).max() > 0.5 and not np.array_equal(ref[0], ref[3])
    # StructTuple round trip with the (devices, E/devices) leading dims of the reference
    params = inference._struct_tuple(net, theta)
    assert params.var0.shape == (2, 3) and params[4].shape == (2, 3) + net.leaves[4].shape
    np.testing.assert_array_equal(inference._flatten_struct(net, params), theta)
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
