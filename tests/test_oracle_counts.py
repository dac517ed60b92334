"""NB / ZINB pieces of the oracle against an INDEPENDENT implementation (scipy.stats.nbinom).

The reference has no golden file for its count models (tests/test_data holds NORMAL runs only), and
TensorFlow Probability cannot be imported here, so `oracle.nb_log_prob / count_forecast / count_cdf`
restate tfd.NegativeBinomial(total_count, logits) from its documented definition: `total_count`
failures, success probability sigmoid(logits), pmf(y) = C(tc + y - 1, y) (1 - p)^tc p^y.  scipy's
nbinom(n, p_scipy) counts failures before n successes with success probability p_scipy, i.e. the same
law with n = total_count and p_scipy = 1 - p = sigmoid(-logits).  These checks pin the closed forms
(log-pmf, mean, variance, cdf, zero inflation) to that library; the mapping network output ->
(total_count, logits) follows the reference's code (models.py:166-191) and is covered by the
finite-difference and GPU parity tests.
"""
import numpy as np
from scipy import stats

from oracle import bnf_oracle as O


def _params(seed=0, E=3, n=40):
  rng = np.random.default_rng(seed)
  tc = np.exp(rng.uniform(-1.5, 2.0, E))                 # total_count in (0.2, 7.4): non-integer on purpose
  logits = rng.uniform(-3.0, 3.0, (E, n))
  y = rng.integers(0, 60, (E, n)).astype(np.float64)
  return tc, logits, y


def test_nb_log_prob_matches_scipy():
  tc, logits, y = _params()
  got = O.nb_log_prob(y, tc, logits)
  ref = stats.nbinom.logpmf(y, tc[:, None], O.sigmoid(-logits))
  np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)


def test_zinb_log_prob_matches_mixture_of_scipy():
  tc, logits, y = _params(1)
  y[:, ::3] = 0.0
  pi = np.array([0.05, 0.3, 0.7])[:, None]
  got = O.zinb_log_prob(y, tc, logits, pi)
  pmf = stats.nbinom.pmf(y, tc[:, None], O.sigmoid(-logits))
  ref = np.log((1 - pi) * pmf + pi * (y == 0))
  np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12)


def test_count_moments_and_cdf_match_scipy():
  tc, logits, _ = _params(2)
  p_scipy = O.sigmoid(-logits)
  mean = tc[:, None] * np.exp(logits)
  np.testing.assert_allclose(mean, stats.nbinom.mean(tc[:, None], p_scipy), rtol=1e-10)
  np.testing.assert_allclose(mean / O.sigmoid(-logits), stats.nbinom.var(tc[:, None], p_scipy), rtol=1e-10)
  fc = dict(tc=tc[:, None], logits=logits, pi=None)
  for x in (0.0, 1.0, 7.0, 33.0):
    np.testing.assert_allclose(O.count_cdf(fc, np.full((1, logits.shape[1]), x)),
                               stats.nbinom.cdf(x, tc[:, None], p_scipy), rtol=1e-9, atol=1e-12)
  pi = np.array([0.1, 0.4, 0.8])[:, None]
  fz = dict(tc=tc[:, None], logits=logits, pi=pi)
  x = 5.0
  np.testing.assert_allclose(O.count_cdf(fz, np.full((1, logits.shape[1]), x)),
                             pi + (1 - pi) * stats.nbinom.cdf(x, tc[:, None], p_scipy), rtol=1e-9)


def test_count_forecast_moments_of_the_mixture():
  """count_forecast's ZINB mean / stddev are those of Mixture([1 - pi, pi], [NB, delta_0])."""
  from tests import util
  net, model, X, y = util.make_problem(n_rows=30, width=64, depth=1, observation_model='ZINB')
  theta = util.random_theta(model, 2, scale=0.3)
  out = O.forward(model, theta, X)
  fc = O.count_forecast(model, theta, out)
  p_scipy = O.sigmoid(-fc['logits'])
  m_nb = stats.nbinom.mean(fc['tc'], p_scipy)
  v_nb = stats.nbinom.var(fc['tc'], p_scipy)
  pi = fc['pi']
  mean = (1 - pi) * m_nb
  var = (1 - pi) * (v_nb + m_nb ** 2) - mean ** 2
  np.testing.assert_allclose(fc['mean'], mean, rtol=1e-9)
  np.testing.assert_allclose(fc['stddev'], np.sqrt(var), rtol=1e-9)
