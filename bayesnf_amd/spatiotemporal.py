"""Estimator API of the MI355X-native BayesNF path.

Drop-in for the public surface of /root/reference/src/bayesnf/spatiotemporal.py:
`BayesianNeuralFieldMAP`, `BayesianNeuralFieldMLE`, `BayesianNeuralFieldVI`
(ctor kwargs :217-232, `MAP.fit` :480-489, `VI.fit` :565-576, `predict`
:372-408, attributes `params_`, `losses_`, `data_handler`) and of the pandas
glue (`SpatiotemporalDataHandler` :114-192, `seasonality_to_float` :31-59,
`seasonalities_to_array` :62-95).  The three engine calls underneath
(`inference.fit_map / fit_vi / predict_bnf`, reference seam :400,:529,:634) go
to the HIP library instead of jax.

Differences a caller can see:
  * `seed` is an int or a length-2 uint32 array (a `jax.random.PRNGKey` also
    works, it is such an array).  With `init_rng='jax'` (the default) every
    estimator draws from the reference's own streams for that seed (threefry + the
    TFP seed chain, `jaxseed`): the initial particles of MAP / MLE and their
    per-member per-epoch minibatch shuffles (`jax.random.permutation`); the initial
    surrogate means, the optimisation noise and the posterior draws of a full-batch
    VI fit -- full-batch fits reproduce the reference's golden predictions.  Only a
    VI fit WITH `batch_size` keeps this package's counter-based generator for its
    noise and row batches (which split of the step seed the reference uses there is
    pinned by no golden).  `init_rng='philox'` selects the device generator everywhere.
  * limits of the HIP engine, checked when the estimator is constructed / fitted
    (the reference accepts any size): `depth` <= 8 (any `width` up to 8192 -- widths that are
    not a multiple of 64 run zero-padded inside the engine, same model and parameters);
    at most 8 feature columns, 96 distinct seasonal frequencies, 16 interactions;
    at most 512 (fp32: 256) features in total; `batch_size` <= number of rows.
  * the leading `(num_devices, ensemble_size // num_devices)` dimensions use
    `num_devices = distributed.device_count()`: the devices this ONE process drives
    (every visible GPU, or `BNF_DEVICES`; the reference's `jax.local_devices()` shape),
    or the torch.distributed world size under a one-process-per-GPU launcher.
  * `likelihood_model()` returns `bayesnf_amd.inference.EnsembleLikelihood`,
    not a TFP distribution.
"""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np
import pandas as pd

from . import distributed
from . import inference


# ---------------------------------------------------------------------------
# pandas helpers
# ---------------------------------------------------------------------------
def seasonality_to_float(seasonality: str, freq: str) -> float:
  """How many `freq` steps make one `seasonality` period, on average.

  Averaged over the four years 2020-2023 so that a leap day is included, e.g.
  ('Y','D') -> 365.25, ('M','D') -> 30.4375, ('M','h') -> 730.5.
  """
  anchors = pd.date_range('2020-01-01', periods=5, freq='YS')
  coarse = anchors.to_period(seasonality)
  n_coarse = (coarse[-1] - coarse[0]).n
  fine = pd.date_range(coarse[0].start_time,
                       coarse[-1].start_time).to_period(freq)
  n_fine = (fine[-1] - fine[0]).n
  return n_fine / n_coarse


def seasonalities_to_array(seasonalities: Sequence[float | str],
                           freq: str) -> np.ndarray:
  """Periods (numbers, or pandas offset aliases) as floats in units of `freq`.

  Raises TypeError for a period shorter than one `freq` step.
  """
  out = []
  for s in seasonalities:
    if isinstance(s, str):
      value = seasonality_to_float(s, freq)
      if value < 1:
        raise TypeError(
            f'seasonality={s!r} should represent a time span greater than '
            f'freq={freq!r}, but {s} is {value:.2f} of a {freq}')
    else:
      value = s
      if value < 1:
        raise TypeError(f'seasonality_float={value!r} should be larger than 1.')
    out.append(value)
  return np.array(out)


_EPOCH = '2020-01-01'


def _convert_datetime_col(table, time_column, timetype, freq, time_min=None):
  """Replace `time_column` by a number, shifted so the training minimum is 0.

  'index': number of `freq` periods since 2020-01-01; 'float': the value as is.
  Mutates `table` (callers pass a copy) and returns (table, time_min).
  """
  col = table[time_column]
  if timetype == 'index':
    origin = pd.to_datetime(_EPOCH).to_period(freq)
    col = (col.dt.to_period(freq) - origin).apply(lambda delta: delta.n)
  elif timetype == 'float':
    col = col.apply(float)
  else:
    raise ValueError(f'Unknown timetype: {timetype}')
  if time_min is None:
    time_min = col.min()
  table[time_column] = col - time_min
  return table, time_min


class SpatiotemporalDataHandler:
  """DataFrame -> (N, D) float features / (N,) target.

  Column 0 of `feature_cols` is time.  `get_train` learns the time origin, the
  time scale (max training index, used as `input_scales[0]`) and the
  mean / std of the columns listed in `standardize`; `get_test` re-applies them.
  """

  def __init__(self, feature_cols, target_col, timetype, freq,
               standardize=None):
    self.feature_cols = feature_cols
    self.target_col = target_col
    self.timetype = timetype
    self.freq = freq
    self.standardize = standardize
    self.mu_ = None
    self.std_ = None
    self.time_min_ = None
    self.time_scale_ = None

  @property
  def _time_idx(self) -> int:
    return 0

  @property
  def _time_column(self) -> str:
    return self.feature_cols[self._time_idx]

  def _maybe_filter_target_nans(self, table):
    if self.target_col in table.columns:
      return table[table[self.target_col].notna()]
    return table

  def copy_and_filter_table(self, table):
    return self._maybe_filter_target_nans(table.copy())

  def get_target(self, table) -> np.ndarray:
    return self._maybe_filter_target_nans(table)[self.target_col].values

  def get_train(self, table) -> np.ndarray:
    work = self.copy_and_filter_table(table)
    ncol = len(self.feature_cols)
    self.mu_, self.std_ = np.zeros(ncol), np.ones(ncol)
    work, self.time_min_ = _convert_datetime_col(
        work, self._time_column, self.timetype, self.freq, None)
    feats = work[self.feature_cols].values
    self.time_scale_ = feats[:, self._time_idx].max()
    if self.standardize:
      if self._time_column in self.standardize:
        raise TypeError('Do not standardize the time column!')
      cols = [self.feature_cols.index(c) for c in self.standardize]
      block = feats[:, cols].astype(float)
      self.mu_[cols] = block.mean(axis=0)
      self.std_[cols] = block.std(axis=0)
      feats = (feats - self.mu_) / self.std_
    return feats

  def get_test(self, table) -> np.ndarray:
    work, _ = _convert_datetime_col(
        table.copy(), self._time_column, self.timetype, self.freq,
        self.time_min_)
    feats = work[self.feature_cols].values
    if self.standardize:
      feats = (feats - self.mu_) / self.std_
    return feats

  def get_input_scales(self) -> np.ndarray:
    scales = np.ones(len(self.feature_cols))
    scales[self._time_idx] = self.time_scale_
    return scales


# ---------------------------------------------------------------------------
# estimators
# ---------------------------------------------------------------------------
class BayesianNeuralFieldEstimator:
  """Common constructor / predict for the MAP, MLE and VI estimators.

  Keyword arguments (all keyword-only, as in the reference):
    feature_cols, target_col: column names; feature_cols[0] is the time column.
    seasonality_periods / num_seasonal_harmonics: seasonal Fourier features of
      the raw time index; periods may be pandas aliases when timetype='index'.
    fourier_degrees: octaves of Fourier features per input (default 5 each).
    interactions: list of (i, j) input-column pairs whose product is a feature.
    freq, timetype: 'index' needs a datetime column and `freq`; 'float' a
      float column and no `freq`.
    depth, width: hidden layers and units.
    observation_model: 'NORMAL', 'NB' or 'ZINB'.
    standardize: columns to z-score (never the time column).
  Extra (not in the reference): compute_dtype 'fp32' | 'fp32_split' | 'bf16' | 'fp8' selects the
    arithmetic of the dense contractions on the GPU (fp32 accumulate either
    way; engine.default_dtype); default from env BNF_DTYPE, else 'fp32_split' (f32 storage, split-bf16 contractions:
    within the fp32 parity gates; 'fp32' = the exact f32 MFMA chain).  init_rng 'jax' | 'philox': 'jax'
    (default, env BNF_INIT_RNG) draws the initial Dense kernels from the reference's own streams
    for `seed` (jax threefry + TFP seed chain restated in `jaxseed`), so a full-batch fit follows
    the reference's trajectory; 'philox' uses the device generator.
  """

  _ensemble_dims: int
  _prior_weight: float = 1.0
  _scale_epochs_by_batch_size: bool = False

  def __init__(self, *, feature_cols, target_col, seasonality_periods=None,
               num_seasonal_harmonics=None, fourier_degrees=None,
               interactions=None, freq=None, timetype='index', depth=2,
               width=512, observation_model='NORMAL', standardize=None,
               compute_dtype=None, init_rng=None):
    self.feature_cols = feature_cols
    self.target_col = target_col
    self.seasonality_periods = seasonality_periods
    self.num_seasonal_harmonics = num_seasonal_harmonics
    self.fourier_degrees = fourier_degrees
    self.interactions = interactions
    self.freq = freq
    self.timetype = timetype
    self.depth = depth
    self.width = width
    self.observation_model = observation_model
    self.standardize = standardize
    self.compute_dtype = compute_dtype
    self.init_rng = init_rng
    self._check_engine_limits(len(feature_cols))
    self.losses_ = None
    self.params_ = None
    self.data_handler = SpatiotemporalDataHandler(
        feature_cols, target_col, timetype, freq, standardize=standardize)

  # ---- argument normalisation (reference :296-358) ------------------------
  def _get_fourier_degrees(self, batch_shape):
    ncol = batch_shape[-1]
    if self.fourier_degrees is None:
      return np.full(ncol, 5, dtype=int)
    degrees = np.atleast_1d(self.fourier_degrees).astype(int)
    if degrees.shape[-1] != ncol:
      raise ValueError(
          f'The length of fourier_degrees ({degrees.shape[-1]}) must match the '
          f'input dimension dimension ({ncol}).')
    return degrees

  def _get_interactions(self):
    if self.interactions is None:
      return np.zeros((0, 2), dtype=int)
    pairs = np.array(self.interactions).astype(int)
    if pairs.ndim != 2 or pairs.shape[-1] != 2:
      raise ValueError(
          'The argument for `interactions` should be a 2-d array of integers '
          'of shape (N, 2), indicating the column indices to interact (the '
          f' passed shape was {pairs.shape})')
    return pairs

  def _get_seasonality_periods(self):
    index_time = self.timetype == 'index'
    if (index_time and self.freq is None) or (
        self.timetype == 'float' and self.freq is not None):
      raise ValueError(f'Invalid {self.freq=} with {self.timetype=}.')
    if self.seasonality_periods is None:
      return np.zeros(0)
    if index_time:
      return seasonalities_to_array(self.seasonality_periods, self.freq)
    if self.timetype == 'float':
      return np.asarray(self.seasonality_periods, dtype=float)
    raise AssertionError(f'Impossible {self.timetype=}.')

  def _get_num_seasonal_harmonics(self):
    if self.timetype == 'index':
      if self.num_seasonal_harmonics is None:
        return np.zeros(0)
      return np.array(self.num_seasonal_harmonics)
    if self.timetype == 'float':
      if self.num_seasonal_harmonics is not None:
        raise ValueError(
            f'Cannot use num_seasonal_harmonics with {self.timetype=}.')
      # continuous time: exactly one harmonic per period; any 0 < h <= p/2
      # below 1 makes arange(1, 1 + h) == [1] in the frequency table.
      return np.fmin(.5, self._get_seasonality_periods() / 2)
    raise AssertionError(f'Impossible {self.timetype=}.')

  def _check_engine_limits(self, n_inputs=None):
    """The HIP engine's hard limits, reported here with the estimator's own argument names
    instead of as an error code from deep inside `fit` (include/bnf.h BNF_MAX_*)."""
    if not isinstance(self.width, (int, np.integer)) or not 1 <= self.width <= 8192:
      raise ValueError(f'width={self.width}: the MI355X engine supports widths 1..8192')
    if not 1 <= int(self.depth) <= 8:
      raise ValueError(f'depth={self.depth}: the MI355X engine supports 1..8 hidden layers')
    if n_inputs is not None and n_inputs > 8:
      raise ValueError(f'{n_inputs} feature columns: the MI355X engine supports at most 8')
    if self.interactions is not None and len(self.interactions) > 16:
      raise ValueError(f'{len(self.interactions)} interactions: the MI355X engine supports at most 16')

  def _model_args(self, batch_shape):
    self._check_engine_limits(batch_shape[-1])
    return dict(
        depth=self.depth,
        input_scales=self.data_handler.get_input_scales(),
        num_seasonal_harmonics=self._get_num_seasonal_harmonics(),
        seasonality_periods=self._get_seasonality_periods(),
        width=self.width,
        init_x=batch_shape,
        fourier_degrees=self._get_fourier_degrees(batch_shape),
        interactions=self._get_interactions(),
    )

  # ---- public API -----------------------------------------------------------
  def fit(self, table, seed):
    raise NotImplementedError('Should be implemented by subclass')

  def predict(self, table, quantiles=(0.5,), approximate_quantiles=False):
    """-> (means, quantiles): means has shape
    (num_devices, ensemble_size // num_devices, len(table)) (VI: an extra
    posterior-sample axis after num_devices); quantiles is a list with one
    (len(table),) array per requested level, for the equal-weight mixture of
    all members.  Exact quantiles use Chandrupatla root finding; approximate
    ones the moment-matched Normal."""
    rows = self.data_handler.get_test(table)
    return inference.predict_bnf(
        rows,
        self.observation_model,
        params=self.params_,
        model_args=self._model_args(rows.shape),
        quantiles=quantiles,
        ensemble_dims=self._ensemble_dims,
        approximate_quantiles=approximate_quantiles,
        compute_dtype=self.compute_dtype,
    )

  def likelihood_model(self, table):
    """Predictive distribution of every member at the rows of `table`
    (reference :433-468 returns a TFP Independent(Normal/NB/ZINB))."""
    rows = self.data_handler.get_test(table)
    return inference.likelihood_model(
        rows, self.observation_model, self.params_,
        self._model_args(rows.shape), ensemble_dims=self._ensemble_dims,
        compute_dtype=self.compute_dtype)


class BayesianNeuralFieldMAP(BayesianNeuralFieldEstimator):
  """Ensemble of maximum-a-posteriori fits (Adam on -log posterior)."""

  _ensemble_dims = 2

  def fit(self, table, seed, ensemble_size=16, learning_rate=0.005,
          num_epochs=5_000, batch_size=None, num_splits=1):
    """Train `ensemble_size` independent members; members are sharded over the
    GPUs (ranks) of the job.  `batch_size=None` is full batch; otherwise each
    epoch does len(table)//batch_size Adam steps on a per-member shuffle.
    `num_splits` trains the ensemble in that many sequential chunks."""
    if ensemble_size < distributed.device_count():
      raise ValueError('ensemble_size cannot be smaller than device_count. '
                       'https://github.com/google/bayesnf/issues/28.')
    x = self.data_handler.get_train(table)
    y = self.data_handler.get_target(table)
    if batch_size is None:
      batch_size = x.shape[0]
    if self._scale_epochs_by_batch_size:
      num_epochs = num_epochs * (x.shape[0] // batch_size)
    self.params_, self.losses_ = inference.fit_map(
        x, y,
        seed=seed,
        observation_model=self.observation_model,
        model_args=self._model_args((batch_size, x.shape[-1])),
        num_particles=ensemble_size,
        learning_rate=learning_rate,
        num_epochs=num_epochs,
        prior_weight=self._prior_weight,
        batch_size=batch_size,
        num_splits=num_splits,
        compute_dtype=self.compute_dtype,
        init_rng=self.init_rng)
    return self


class BayesianNeuralFieldMLE(BayesianNeuralFieldMAP):
  """Ensemble of maximum-likelihood fits: MAP without the prior term."""

  _prior_weight = 0.0


class BayesianNeuralFieldVI(BayesianNeuralFieldEstimator):
  """Ensemble of mean-field Gaussian surrogate posteriors (reparameterised
  ELBO with the KL term weighted by `kl_weight`)."""

  _ensemble_dims = 3
  _scale_epochs_by_batch_size = True

  def fit(self, table, seed, ensemble_size=16, learning_rate=0.01,
          num_epochs=1_000, sample_size_posterior=30, sample_size_divergence=5,
          kl_weight=0.1, batch_size=None):
    """`num_epochs` is multiplied by len(table)//batch_size to get the number
    of optimisation steps; each step uses one random batch.  After fitting,
    `sample_size_posterior` parameter draws per member are kept as `params_`."""
    x = self.data_handler.get_train(table)
    y = self.data_handler.get_target(table)
    if batch_size is None:
      batch_size = x.shape[0]
    if self._scale_epochs_by_batch_size:
      num_epochs = num_epochs * (x.shape[0] // batch_size)
    _, self.losses_, self.params_ = inference.fit_vi(
        x, y,
        seed=seed,
        observation_model=self.observation_model,
        model_args=self._model_args((batch_size, x.shape[-1])),
        ensemble_size=ensemble_size,
        learning_rate=learning_rate,
        num_epochs=num_epochs,
        sample_size_posterior=sample_size_posterior,
        sample_size_divergence=sample_size_divergence,
        kl_weight=kl_weight,
        batch_size=batch_size,
        compute_dtype=self.compute_dtype,
        init_rng=self.init_rng)
    return self
