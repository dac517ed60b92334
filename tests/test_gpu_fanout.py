"""One process, several device handles: the reference's own multi-device shape (`jax.pmap` over
`jax.local_devices()`, /root/reference/src/bayesnf/inference.py:573-579,445; result shapes
spatiotemporal.py:389-392; ValueError :519-521).  A 1-GPU box runs it with the device list
BNF_DEVICES=0,0 (two handles on one GPU, driven from two host threads): member g * E/G + k of the
fan-out must equal member g * E/G + k of the one-handle fit, and predictions must not depend on the
device count the parameters were fitted or are predicted on."""
import os

import numpy as np
import pandas as pd
import pytest

from bayesnf_amd import BayesianNeuralFieldMAP, BayesianNeuralFieldVI, distributed

pytestmark = pytest.mark.gpu

MODEL = dict(width=64, depth=2, seasonality_periods=np.asarray([4.0, 52.1775]),
             num_seasonal_harmonics=np.asarray([2.0, 4]), observation_model='NORMAL',
             feature_cols=['datetime', 'latitude', 'longitude'], target_col='chickenpox',
             timetype='index', freq='W', standardize=['latitude', 'longitude'])


def _frame(golden_dir):
  return pd.read_csv(os.path.join(golden_dir, 'chickenpox.8.train.csv'), index_col=0, parse_dates=['datetime'])


def _flat(a):
  """(devices, members / devices, ...) -> (members, ...)"""
  a = np.asarray(a)
  return a.reshape((a.shape[0] * a.shape[1],) + a.shape[2:])


@pytest.mark.parametrize('batch_size', [None, 32])
def test_map_fanout_equals_one_handle(golden_dir, monkeypatch, batch_size):
  df = _frame(golden_dir)
  kw = dict(seed=3, ensemble_size=6, num_epochs=6, learning_rate=0.01, batch_size=batch_size)
  monkeypatch.setenv('BNF_DEVICES', '0')
  one = BayesianNeuralFieldMAP(**MODEL).fit(df, **kw)
  m_one, q_one = one.predict(df, quantiles=(0.5, 0.9))
  monkeypatch.setenv('BNF_DEVICES', '0,0,0')
  assert distributed.device_count() == 3 and [s.index for s in distributed.local_shards()] == [0, 1, 2]
  fan = BayesianNeuralFieldMAP(**MODEL).fit(df, **kw)
  assert fan.losses_.shape == (3, 2, 6) and one.losses_.shape == (1, 6, 6)
  assert fan.params_.var4.shape == (3, 2) + one.params_.var4.shape[2:]
  np.testing.assert_allclose(_flat(fan.losses_), _flat(one.losses_), rtol=1e-5)
  for a, b in zip(fan.params_, one.params_):
    np.testing.assert_allclose(_flat(a), _flat(b), rtol=2e-5, atol=2e-6)
  m_fan, q_fan = fan.predict(df, quantiles=(0.5, 0.9))
  assert m_fan.shape == (3, 2, 100)
  np.testing.assert_allclose(_flat(m_fan), _flat(m_one), rtol=1e-4, atol=1e-5)
  np.testing.assert_allclose(q_fan[1], q_one[1], rtol=1e-4, atol=1e-4)
  # parameters fitted on three devices, predicted on one and on two (6 members -> 3 + 3): same numbers
  for devs in ('0', '0,0'):
    monkeypatch.setenv('BNF_DEVICES', devs)
    m_x, q_x = fan.predict(df, quantiles=(0.5, 0.9))
    assert m_x.shape == (3, 2, 100)
    np.testing.assert_allclose(m_x, m_fan, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(q_x[0], q_fan[0], rtol=1e-5, atol=1e-5)
  # five members over two devices at predict time: an uneven deal (3 + 2, the short block padded)
  monkeypatch.setenv('BNF_DEVICES', '0')
  five = BayesianNeuralFieldMAP(**MODEL).fit(df, seed=3, ensemble_size=5, num_epochs=2)
  m5, _ = five.predict(df)
  monkeypatch.setenv('BNF_DEVICES', '0,0')
  m5b, _ = five.predict(df)
  np.testing.assert_allclose(m5b, m5, rtol=1e-6, atol=1e-6)


def test_vi_fanout_equals_one_handle(golden_dir, monkeypatch):
  """With the engine's own generator (keyed by the GLOBAL member id) a VI fit does not depend on how
  the members are dealt out.  (The reference's VI noise stream does: `split(fit_seed, devices)[rank]`,
  inference.py:727-738 -- under init_rng='jax' a fan-out follows the reference at THAT device count.)"""
  monkeypatch.setenv('BNF_INIT_RNG', 'philox')
  df = _frame(golden_dir)
  kw = dict(seed=5, ensemble_size=4, num_epochs=4, learning_rate=0.01, sample_size_posterior=3,
            sample_size_divergence=2, kl_weight=0.1)
  monkeypatch.setenv('BNF_DEVICES', '0')
  one = BayesianNeuralFieldVI(**MODEL).fit(df, **kw)
  monkeypatch.setenv('BNF_DEVICES', '0,0')
  fan = BayesianNeuralFieldVI(**MODEL).fit(df, **kw)
  assert fan.losses_.shape == (2, 2, 4)
  np.testing.assert_allclose(_flat(fan.losses_), _flat(one.losses_), rtol=2e-5)
  m_one, _ = one.predict(df)
  m_fan, _ = fan.predict(df)
  assert m_fan.shape[0] == 2 and m_one.shape[0] == 1
  # (devices, draws, members / devices, rows): compare draw by draw, member by member
  a = np.moveaxis(m_fan, 1, 0).reshape(m_fan.shape[1], -1, m_fan.shape[-1])
  b = np.moveaxis(m_one, 1, 0).reshape(m_one.shape[1], -1, m_one.shape[-1])
  np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5)


def test_fewer_members_than_devices_is_rejected(golden_dir, monkeypatch):
  monkeypatch.setenv('BNF_DEVICES', '0,0,0,0')
  with pytest.raises(ValueError, match='ensemble_size'):
    BayesianNeuralFieldMAP(**MODEL).fit(_frame(golden_dir), seed=0, ensemble_size=3, num_epochs=1)
