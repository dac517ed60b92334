"""Two real ranks on two real GPUs (skipped, not failed, on 1-GPU boxes): one process per GPU under
torch.distributed.run, backend nccl = RCCL.  (1) every rank fits its member shard through the public
estimator, the fitted parameters are all-gathered, and the result equals the one-process fit member for
member; (2) bench.py's own 2-rank path with the posterior gather through the C ABI (`bnf_allgather`)."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (one rank per GPU)')]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, pandas as pd, torch
    sys.path.insert(0, %(root)r)
    from bayesnf_amd import BayesianNeuralFieldMAP, distributed
    distributed.maybe_init_from_env()
    assert distributed.is_distributed() and distributed.device_count() == 2
    df = pd.read_csv(os.path.join(%(root)r, 'tests', 'golden', 'chickenpox.8.train.csv'), index_col=0, parse_dates=['datetime'])
    est = BayesianNeuralFieldMAP(width=64, depth=2, seasonality_periods=np.asarray([4.0, 52.1775]),
                                 num_seasonal_harmonics=np.asarray([2.0, 4]), observation_model='NORMAL',
                                 feature_cols=['datetime', 'latitude', 'longitude'], target_col='chickenpox',
                                 timetype='index', freq='W', standardize=['latitude', 'longitude'])
    est.fit(df, seed=3, ensemble_size=6, num_epochs=6, learning_rate=0.01)
    means, qs = est.predict(df, quantiles=(0.5,))
    if distributed.rank() == 0:
      np.savez(%(out)r, losses=est.losses_, var4=est.params_.var4, means=means, q50=qs[0])
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
''')


def _env():
  env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'BNF_DEVICES')}
  env.update(MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  return env


def test_two_nccl_ranks_fit_equals_one_process_fit(tmp_path, golden_dir, monkeypatch):
  out = str(tmp_path / 'two_rank.npz')
  script = tmp_path / 'worker.py'
  script.write_text(WORKER % dict(root=ROOT, out=out))
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', '29541', str(script)]
  for gather in ('torch', 'cabi'):
    r = subprocess.run(cmd, env=dict(_env(), BNF_GATHER=gather), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    two = np.load(out)
    import pandas as pd
    from bayesnf_amd import BayesianNeuralFieldMAP
    monkeypatch.setenv('BNF_DEVICES', '0')
    df = pd.read_csv(os.path.join(golden_dir, 'chickenpox.8.train.csv'), index_col=0, parse_dates=['datetime'])
    one = BayesianNeuralFieldMAP(width=64, depth=2, seasonality_periods=np.asarray([4.0, 52.1775]),
                                 num_seasonal_harmonics=np.asarray([2.0, 4]), observation_model='NORMAL',
                                 feature_cols=['datetime', 'latitude', 'longitude'], target_col='chickenpox',
                                 timetype='index', freq='W', standardize=['latitude', 'longitude'])
    one.fit(df, seed=3, ensemble_size=6, num_epochs=6, learning_rate=0.01)
    m1, q1 = one.predict(df, quantiles=(0.5,))
    assert two['losses'].shape == (2, 3, 6) and one.losses_.shape == (1, 6, 6)
    np.testing.assert_allclose(two['losses'].reshape(6, 6), one.losses_.reshape(6, 6), rtol=1e-5)
    np.testing.assert_allclose(two['var4'].reshape((6,) + two['var4'].shape[2:]),
                               np.asarray(one.params_.var4).reshape((6,) + two['var4'].shape[2:]), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(two['means'].reshape(6, -1), m1.reshape(6, -1), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(two['q50'], q1[0], rtol=1e-4, atol=1e-4)


def test_bench_two_ranks_gathers_through_the_c_abi():
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                      '--members-per-gpu', '8', '--no-cpu-baseline'], env=_env(), capture_output=True, text=True,
                     timeout=900, cwd=ROOT)
  assert r.returncode == 0, r.stderr[-3000:]
  d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
  g = d['posterior_gather']
  assert d['n_gpus'] == 2 and d['rccl_world_size'] == 2 and 'cabi_error' not in g, g
  assert 'bnf_allgather' in g['impl'] and g['rank_checksums_ok'] and g['finite'] and g['shape'] == [16, 1024]


def test_one_process_two_gpus_gathers_through_the_rccl_group(monkeypatch, golden_dir):
  """The reference's own shape on two real GPUs: ONE process, one engine handle + host thread per device
  (`distributed.run_shards`), the fitted parameters / predictive means assembled by one grouped RCCL all-gather over
  the local communicator set (bnf_comm_create_local + bnf_allgather_group) -- equal to the one-device fit member for
  member, and to the same gather done with peer copies."""
  import pandas as pd
  from bayesnf_amd import BayesianNeuralFieldMAP, distributed
  df = pd.read_csv(os.path.join(golden_dir, 'chickenpox.8.train.csv'), index_col=0, parse_dates=['datetime'])
  def fit(devs, gather):
    monkeypatch.setenv('BNF_DEVICES', devs)
    monkeypatch.setenv('BNF_GATHER', gather)
    est = BayesianNeuralFieldMAP(width=64, depth=2, seasonality_periods=np.asarray([4.0, 52.1775]),
                                 num_seasonal_harmonics=np.asarray([2.0, 4]), observation_model='NORMAL',
                                 feature_cols=['datetime', 'latitude', 'longitude'], target_col='chickenpox',
                                 timetype='index', freq='W', standardize=['latitude', 'longitude'])
    est.fit(df, seed=3, ensemble_size=6, num_epochs=6, learning_rate=0.01)
    means, _ = est.predict(df, quantiles=(0.5,))
    return est.losses_.reshape(6, -1), means.reshape(6, -1), distributed.last_gather()
  l2, m2, note = fit('0,1', 'rccl')
  assert note.get('impl') == 'rccl-group', note
  lp, mp, notep = fit('0,1', 'peer')
  assert notep.get('impl') == 'peer-copies'
  l1, m1, _ = fit('0', 'rccl')
  # the DEFAULT (round 6): the collective once the device set has passed its time-limited first-use check
  distributed._group_verdict.clear()
  la, ma, notea = fit('0,1', 'auto')
  assert notea.get('impl') == 'rccl-group' and notea['check']['ok'] is True, notea
  np.testing.assert_array_equal(la, l2)
  np.testing.assert_array_equal(ma, m2)
  np.testing.assert_array_equal(l2, lp)
  np.testing.assert_array_equal(m2, mp)
  np.testing.assert_allclose(l2, l1, rtol=1e-5)
  np.testing.assert_allclose(m2, m1, rtol=1e-4, atol=1e-5)


def test_bench_inproc_launcher_on_two_gpus():
  for extra in (['--check'], ['--steps', '3', '--warmup', '1', '--members-per-gpu', '8', '--no-cpu-baseline']):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launcher', 'inproc'] + extra,
                       env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    if extra[0] == '--check':
      assert d['check'] == 'ok' and d['gather_impl'] == 'rccl-group' and d['devices_visible'] >= 2
    else:
      assert d['n_gpus'] == 2 and d['launcher'].startswith('inproc') and d['rank_devices'] == ['cuda:0', 'cuda:1']
      g = d['posterior_gather']
      assert 'bnf_allgather_group' in g['impl'] and g['rank_checksums_ok'] and g['finite'] and g['shape'] == [16, 1024]
