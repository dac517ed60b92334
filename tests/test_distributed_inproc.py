"""The one-process, several-devices mode on CPU (bayesnf_amd/distributed.py): a real sharded `fit_map` over two shards of
one process (BNF_DEVICES=0,0; stand-in engine behind the one seam, tests/standin_engine.py) equals the one-shard fit, and
the posterior gather's first-use check -- `group_gather_verdict`: a 1 KiB grouped all-gather on a watchdog thread -- makes
the RCCL collective the default when it passes and the peer copies when it fails, returns wrong blocks, or HANGS (a hang
must not stall a fit whose training has finished).  The collective itself is replaced by stand-ins here; the hardware
proof is tests/test_gpu_multirank.py."""
import threading
import time

import numpy as np
import pytest
import torch

from bayesnf_amd import distributed, inference, jaxseed
from bayesnf_amd.spec import NetSpec
from tests.standin_engine import StandInEngine

KW = dict(width=64, depth=1, input_scales=[9.0, 1.0], fourier_degrees=[2, 0], interactions=[[0, 1]],
          seasonality_periods=[], num_seasonal_harmonics=[])


def _problem():
  rng = np.random.default_rng(0)
  X = np.stack([rng.integers(0, 10, 40).astype(float), rng.standard_normal(40)], axis=1)
  y = np.sin(X[:, 0]) + X[:, 1] + 0.1 * rng.standard_normal(40)
  return X, y


@pytest.fixture
def standin(monkeypatch):
  StandInEngine.model_kwargs = dict(KW, observation_model='NORMAL')
  StandInEngine.created.clear()
  monkeypatch.setattr(inference, 'Engine', StandInEngine)
  distributed._group_verdict.clear()
  distributed._gather_note.clear()
  yield StandInEngine
  distributed._group_verdict.clear()


@pytest.mark.parametrize('bs', [None, 16])
def test_two_shards_of_one_process_equal_one_shard(standin, monkeypatch, bs):
  X, y = _problem()
  args = dict(KW, init_x=X[:2])
  out = {}
  for devs in ('0', '0,0'):
    monkeypatch.setenv('BNF_DEVICES', devs)
    standin.created.clear()
    params, losses = inference.fit_map(X, y, 7, 'NORMAL', args, num_particles=6, learning_rate=0.01, num_epochs=3, batch_size=bs)
    G = len(devs.split(','))
    assert sorted(standin.created) == [(g * (6 // G), 6 // G, 0) for g in range(G)]
    assert losses.shape == (G, 6 // G, 3) and params.var0.shape == (G, 6 // G)
    out[devs] = (inference._flatten_struct(NetSpec(**KW), params).reshape(6, -1), losses.reshape(6, 3))
    assert distributed.last_gather().get('impl', 'peer-copies') == 'peer-copies'     # repeated ordinals: never the collective
  np.testing.assert_array_equal(out['0'][0], out['0,0'][0])
  np.testing.assert_array_equal(out['0'][1], out['0,0'][1])


def _copying_allgather(sends, recvs):
  for r in recvs:
    for i, s in enumerate(sends):
      r[i].copy_(s)


def test_group_gather_is_the_default_once_its_first_use_check_passes(standin, monkeypatch):
  monkeypatch.delenv('BNF_GATHER', raising=False)
  monkeypatch.setattr(distributed, '_distinct_gpus', lambda parts: len(parts) > 1)
  calls = []
  monkeypatch.setattr(distributed, '_group_allgather', lambda s, r: (calls.append(tuple(s[0].shape)), _copying_allgather(s, r)))
  parts = [torch.arange(12.).reshape(3, 4) + 100 * g for g in range(2)]
  got = distributed.gather_shards(parts)
  np.testing.assert_array_equal(got.numpy(), torch.stack(parts).numpy())
  note = distributed.last_gather()
  assert note['impl'] == 'rccl-group' and note['check']['ok'] is True
  assert calls == [(256,), (3, 4)]                       # the 1 KiB check, then the payload
  distributed.gather_shards(parts)
  assert calls == [(256,), (3, 4), (3, 4)]               # the verdict is cached per device set
  monkeypatch.setenv('BNF_GATHER', 'peer')               # the escape hatch
  distributed.gather_shards(parts)
  assert distributed.last_gather()['impl'] == 'peer-copies' and len(calls) == 3


@pytest.mark.parametrize('failure', ['hang', 'wrong', 'raise'])
def test_a_failing_or_hanging_collective_falls_back_to_peer_copies(standin, monkeypatch, failure):
  monkeypatch.delenv('BNF_GATHER', raising=False)
  monkeypatch.setenv('BNF_GATHER_TIMEOUT_S', '0.5')
  monkeypatch.setattr(distributed, '_distinct_gpus', lambda parts: len(parts) > 1)
  release = threading.Event()

  def bad(sends, recvs):
    if failure == 'hang':
      release.wait(30)                                    # what a stuck ncclCommInitAll looks like from here
    elif failure == 'wrong':
      for r in recvs:
        r.fill_(7.0)
    else:
      raise RuntimeError('ncclCommInitAll: unhandled system error')
  monkeypatch.setattr(distributed, '_group_allgather', bad)
  parts = [torch.arange(6.) + 10 * g for g in range(3)]
  t0 = time.time()
  got = distributed.gather_shards(parts)
  assert time.time() - t0 < 5.0                           # the fit is not stalled
  np.testing.assert_array_equal(got.numpy(), torch.stack(parts).numpy())
  note = distributed.last_gather()
  assert note['impl'] == 'peer-copies' and note['check']['ok'] is False
  assert {'hang': 'no answer within', 'wrong': 'wrong blocks', 'raise': 'ncclCommInitAll'}[failure] in note['check']['error']
  t0 = time.time()
  distributed.gather_shards(parts)                        # cached: no second wait
  assert time.time() - t0 < 0.3 and distributed.last_gather()['impl'] == 'peer-copies'
  release.set()
