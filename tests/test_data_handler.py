"""Host-side (pandas) behaviour of the estimator layer, CPU only.

Same cases as the reference's unit tests
(/root/reference/tests/test_spatiotemporal.py:21-120: KAT K5/K6 of SURVEY.md)
plus the data-handler and argument-validation paths the reference leaves
untested."""
import numpy as np
import pandas as pd
import pytest

from bayesnf_amd import BayesianNeuralFieldMAP, BayesianNeuralFieldMLE, BayesianNeuralFieldVI
from bayesnf_amd import spatiotemporal as st
from bayesnf_amd.spec import NetSpec, seasonal_frequency_table

K5 = [('Y', 'Y', 1), ('Q', 'Q', 1), ('Y', 'Q', 4), ('M', 'h', 730.5), ('Q', 'M', 3),
      ('Y', 'M', 12), ('M', 'D', 30.4375), ('min', 's', 60), ('h', 's', 3600),
      ('D', 's', 86400), ('M', 's', 2629800), ('Q', 's', 7889400), ('Y', 's', 31557600)]


@pytest.mark.parametrize('seasonality,freq,expected', K5)
def test_seasonality_to_float(seasonality, freq, expected):
  assert st.seasonality_to_float(seasonality, freq) == expected


def test_seasonalities_to_array():
  np.testing.assert_allclose(st.seasonalities_to_array(['D', 'W', 'M'], 'h'), [24, 168, 730.5])
  np.testing.assert_allclose(st.seasonalities_to_array([7, 'Y'], 'D'), [7, 365.25])
  with pytest.raises(TypeError):
    st.seasonalities_to_array(['h'], 'D')
  with pytest.raises(TypeError):
    st.seasonalities_to_array([0.5], 'D')


@pytest.mark.parametrize('p,h', [([], []), ([10, 15], [8, 6])])
def test_periods_harmonics_index_time(p, h):
  m = BayesianNeuralFieldMAP(freq='D', seasonality_periods=p, num_seasonal_harmonics=h,
                             feature_cols=['t'], target_col='x', timetype='index')
  assert np.all(m._get_seasonality_periods() == p)
  assert np.all(m._get_num_seasonal_harmonics() == h)


@pytest.mark.parametrize('p,h', [([], []), ([10, 12, .25], [.5, .5, .125])])
def test_periods_harmonics_float_time(p, h):
  m = BayesianNeuralFieldMAP(seasonality_periods=p, feature_cols=['t'], target_col='x',
                             timetype='float')
  assert np.all(m._get_seasonality_periods() == p)
  assert np.all(m._get_num_seasonal_harmonics() == h)


def test_freq_timetype_mismatch():
  with pytest.raises(ValueError):
    BayesianNeuralFieldMAP(feature_cols=['t'], target_col='x',
                           timetype='index')._get_seasonality_periods()
  with pytest.raises(ValueError):
    BayesianNeuralFieldMAP(freq='M', feature_cols=['t'], target_col='x',
                           timetype='float')._get_seasonality_periods()


def test_string_period_needs_index_time():
  m = BayesianNeuralFieldMAP(seasonality_periods=['W'], feature_cols=['t'], target_col='x',
                             timetype='float')
  with pytest.raises(ValueError):
    m._get_seasonality_periods()


def test_harmonics_not_allowed_with_float_time():
  m = BayesianNeuralFieldMAP(seasonality_periods=[1, 5], num_seasonal_harmonics=[0.5, 1],
                             feature_cols=['t'], target_col='x', timetype='float')
  with pytest.raises(ValueError):
    m._get_num_seasonal_harmonics()


def test_argument_shapes():
  m = BayesianNeuralFieldMAP(feature_cols=['t', 'a'], target_col='x', freq='D',
                             fourier_degrees=[1, 2, 3])
  with pytest.raises(ValueError):
    m._get_fourier_degrees((10, 2))
  assert list(BayesianNeuralFieldMAP(feature_cols=['t', 'a'], target_col='x',
                                     freq='D')._get_fourier_degrees((10, 2))) == [5, 5]
  with pytest.raises(ValueError):
    BayesianNeuralFieldMAP(feature_cols=['t'], target_col='x', freq='D',
                           interactions=[1, 2, 3])._get_interactions()
  assert BayesianNeuralFieldMAP(feature_cols=['t'], target_col='x',
                                freq='D')._get_interactions().shape == (0, 2)


def test_class_constants():
  assert BayesianNeuralFieldMAP._ensemble_dims == 2 and BayesianNeuralFieldMAP._prior_weight == 1.0
  assert BayesianNeuralFieldMLE._prior_weight == 0.0 and BayesianNeuralFieldMLE._ensemble_dims == 2
  assert BayesianNeuralFieldVI._ensemble_dims == 3 and BayesianNeuralFieldVI._scale_epochs_by_batch_size
  with pytest.raises(NotImplementedError):
    st.BayesianNeuralFieldEstimator(feature_cols=['t'], target_col='x').fit(None, 0)


def _frame():
  t = pd.date_range('2021-03-01', periods=6, freq='W-MON')
  return pd.DataFrame({
      'datetime': list(t) * 2,
      'lat': [1.0] * 6 + [3.0] * 6,
      'lon': [10.0] * 6 + [14.0] * 6,
      'y': [1., 2., np.nan, 4., 5., 6., 7., 8., 9., 10., 11., 12.]})


def test_data_handler_roundtrip():
  dh = st.SpatiotemporalDataHandler(['datetime', 'lat', 'lon'], 'y', 'index', 'W',
                                    standardize=['lat', 'lon'])
  df = _frame()
  X = dh.get_train(df)
  y = dh.get_target(df)
  assert X.shape == (11, 3) and y.shape == (11,)          # NaN target row dropped
  assert X[:, 0].min() == 0 and X[:, 0].max() == 5 and dh.time_scale_ == 5
  np.testing.assert_allclose(X[:, 1].mean(), 0, atol=1e-12)
  np.testing.assert_allclose(X[:, 1].std(), 1, atol=1e-12)
  np.testing.assert_array_equal(dh.get_input_scales(), [5, 1, 1])
  Xt = dh.get_test(df)                                    # predict keeps NaN-target rows
  assert Xt.shape == (12, 3)
  later = df.copy()
  later['datetime'] = later['datetime'] + pd.Timedelta(weeks=10)
  assert dh.get_test(later)[:, 0].max() == 15
  assert 'datetime' in df and df['datetime'].dtype.kind == 'M'   # caller's frame untouched


def test_time_column_cannot_be_standardised():
  dh = st.SpatiotemporalDataHandler(['datetime', 'lat'], 'y', 'index', 'W',
                                    standardize=['datetime'])
  with pytest.raises(TypeError):
    dh.get_train(_frame())


def test_float_time():
  dh = st.SpatiotemporalDataHandler(['t'], 'y', 'float', None)
  df = pd.DataFrame({'t': [3, 4, 7], 'y': [1., 2., 3.]})
  np.testing.assert_array_equal(dh.get_train(df)[:, 0], [0, 1, 4])
  with pytest.raises(ValueError):
    st._convert_datetime_col(df.copy(), 't', 'weird', None)


def test_seasonal_frequency_table():
  f, h = seasonal_frequency_table(np.array([4.0, 52.1775]), np.array([2.0, 10]))
  assert f.dtype == np.float32 and len(f) == 12
  np.testing.assert_array_equal(h, [1, 2, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
  # duplicates keep their first occurrence: 12/6 harmonics of periods 12 and 6
  f, h = seasonal_frequency_table(np.array([12.0, 6.0]), np.array([4, 2]))
  np.testing.assert_allclose(f, np.float32([1, 2, 3, 4]) / np.float32(12))
  np.testing.assert_array_equal(h, [1, 2, 3, 4])
  with pytest.raises(ValueError):
    seasonal_frequency_table(np.array([4.0]), np.array([3]))
  with pytest.raises(ValueError):
    seasonal_frequency_table(np.array([4.0, 8.0]), np.array([1]))
  assert seasonal_frequency_table(np.zeros(0), np.zeros(0))[0].size == 0


def test_netspec_matches_oracle_layout():
  from oracle import bnf_oracle as O
  for kw in (dict(width=64, depth=1, input_scales=[9.0], fourier_degrees=[0], interactions=[]),
             dict(width=128, depth=3, input_scales=[99, 1, 1], fourier_degrees=[5, 0, 2],
                  interactions=[(0, 1), (1, 2)], seasonality_periods=[7, 365.25],
                  num_seasonal_harmonics=[3, 10]),
             dict(width=64, depth=11, input_scales=np.ones(8), fourier_degrees=[1] * 8,
                  interactions=[(0, 7)], seasonality_periods=[5.0], num_seasonal_harmonics=[1])):
    a, b = NetSpec(**kw), O.Model(**kw)
    assert (a.F, a.P) == (b.F, b.P)
    assert [(l.name, l.shape, l.offset) for l in a.leaves] == \
           [(l.name, l.shape, l.offset) for l in b.leaves]
    assert [(g.kind, g.arg, g.ncols, g.col0) for g in a.groups] == \
           [({'u': 0, 'fourier': 1, 'seasonal': 2, 'inter': 3}[g[0]], g[1], g[2], g[3])
            for g in b.groups]
    np.testing.assert_array_equal(a.matrix_mask(), b.matrix_mask())
    th = np.arange(2 * a.P, dtype=np.float32).reshape(2, a.P)
    np.testing.assert_array_equal(a.pack(a.unpack(th)), th)
  # sizes quoted in SURVEY.md section 8
  n = NetSpec(width=512, depth=2, input_scales=[1, 1, 1], fourier_degrees=[5, 5, 5],
              interactions=[], seasonality_periods=[4, 52.1775], num_seasonal_harmonics=[2, 10])
  assert (n.F, n.P) == (57, 292880)
  assert n.flops_per_member_step(10232) == pytest.approx(1.79e10, rel=5e-3)


def test_evaluate_tables_match_published_configs():
  """Hyper-parameter tables of the experiment driver (data from the reference's
  scripts/dataset_config.py:77-180 and scripts/evaluate.py:194-302)."""
  from bayesnf_amd import evaluate as ev
  assert sorted(ev.DATASET_CONFIG) == ['air', 'air_quality', 'chickenpox', 'coprecip', 'sst', 'wind']
  assert ev.MODEL_CONFIG['chickenpox']['vi'] == dict(
      width=256, depth=2, seasonality_periods=[4.0, 52.1775], num_seasonal_harmonics=[2.0, 10],
      observation_model='NORMAL')
  assert ev.MODEL_CONFIG['sst']['map']['width'] == 768 and ev.DATASET_CONFIG['sst']['feature_cols'][-1] == 'soi'
  assert ev.INFERENCE_CONFIG['air_quality']['vi'] == dict(
      num_particles=16, num_epochs=500, learning_rate=0.01, batch_size=3500, kl_weight=0.2,
      sample_size_divergence=5)
  assert ev.INFERENCE_CONFIG['wind']['mle'] is ev.INFERENCE_CONFIG['wind']['map']
  assert ev.INFERENCE_CONFIG['sst']['vi']['learning_rate'] == 0.005
  with pytest.raises(ValueError):
    ev.run_experiment('chickenpox', '/nonexistent', '8', '/tmp/x', 'ais', {}, {}, {}, 0)


def test_engine_limits_are_reported_at_construction():
  """Hard limits of the HIP engine surface as ValueErrors naming the estimator argument
  (the reference accepts any depth / column count; see the module docstring of spatiotemporal.py).
  Any width is accepted, like the reference (widths off the 64 grid run zero-padded in the engine)."""
  import pytest as _pytest
  from bayesnf_amd import BayesianNeuralFieldMAP
  kw = dict(feature_cols=['t', 'x'], target_col='y', timetype='float')
  BayesianNeuralFieldMAP(width=100, **kw)
  with _pytest.raises(ValueError, match='width=0'):
    BayesianNeuralFieldMAP(width=0, **kw)
  with _pytest.raises(ValueError, match='depth=9'):
    BayesianNeuralFieldMAP(depth=9, **kw)
  with _pytest.raises(ValueError, match='feature columns'):
    BayesianNeuralFieldMAP(feature_cols=[f'c{i}' for i in range(9)], target_col='y', timetype='float')
  BayesianNeuralFieldMAP(width=192, depth=3, **kw)
