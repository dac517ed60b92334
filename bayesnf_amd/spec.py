"""Static network description: feature columns and the packed parameter vector.

One `NetSpec` is what the reference keeps spread over a flax module
(`BayesianNeuralField1D`, /root/reference/src/bayesnf/models.py:197-273), its
`mlp_template` pytree and the prior's part list
(`prior_model_fn`, models.py:94-103).  Here every ensemble member owns ONE flat
float32 vector of length `P`; leaf `i` of the reference's `params_` tuple
(`var{i}`) is the slice `[offset_i, offset_i + size_i)` reshaped to `shape_i`.

Leaf order = `[log_noise_scale, shape, inflated_loc_probs]` followed by
`jax.tree_util.tree_leaves(mlp_template)`, i.e. the flax variable dict walked
with string-sorted keys: `Dense_0/{bias,kernel}`, `Dense_1/...`,
`feature_inv_sp_scale{i}`, `inv_sp_layer_scale{l}`, `inv_sp_output_scale`,
`log_scale_adjustment`, `logit_activation_weight`.

The same object is serialised into the C `bnf_config` struct (include/bnf.h).
"""

from __future__ import annotations

import collections
from typing import NamedTuple, Sequence

import numpy as np

OBS_MODELS = ('NORMAL', 'NB', 'ZINB')

# feature-group kinds (shared with csrc/bnf_types.h)
GROUP_INPUT = 0      # scaled inputs u_d
GROUP_FOURIER = 1    # cos|sin(2 pi 2^k u_d)/(k+1)
GROUP_SEASONAL = 2   # cos|sin(2 pi f_j t)/h_j
GROUP_INTERACT = 3   # u_p * u_q


class LeafSpec(NamedTuple):
  name: str
  shape: tuple
  offset: int
  size: int


class GroupSpec(NamedTuple):
  kind: int
  arg: int          # input column for GROUP_FOURIER, else -1
  ncols: int
  col0: int         # first feature column of the group
  scale_offset: int  # offset of its feature_inv_sp_scale leaf in the flat vector


def seasonal_frequency_table(seasonality_periods, num_harmonics):
  """Distinct seasonal frequencies h/p and their harmonic numbers.

  float32 arithmetic and first-occurrence ordering as
  `make_seasonal_frequencies` (models.py:36-59); same three ValueErrors.
  """
  p32 = np.array(seasonality_periods, dtype=np.float32)
  nh = np.asarray(num_harmonics)
  if np.any(nh > p32 / 2):
    raise ValueError('Harmonic cannot exceed half seasonal period.')
  if p32.shape != nh.shape:
    raise ValueError('Number of seasonal periods and harmonics must be equal.')
  if nh.ndim != 1:
    raise ValueError(
        'Arguments `num_harmonics` and `seasonality_periods` must be rank 1.')
  if p32.size == 0:
    return np.zeros(0, np.float32), np.zeros(0, np.float32)
  per_period = [np.arange(1, h + 1, dtype=np.float32) for h in nh]
  all_f = np.concatenate([h / p for h, p in zip(per_period, p32)])
  all_h = np.concatenate(per_period)
  first = np.sort(np.unique(all_f, return_index=True)[1])
  return all_f[first].astype(np.float32), all_h[first].astype(np.float32)


class NetSpec:
  """Feature layout + parameter packing for one (width, depth, featuriser)."""

  def __init__(self, *, width, depth, input_scales, fourier_degrees,
               interactions, seasonality_periods=(), num_seasonal_harmonics=(),
               observation_model='NORMAL', init_x=None):
    del init_x  # only the feature count matters here
    if observation_model not in OBS_MODELS:
      raise AssertionError(('Unknown likelihood distribution:',
                            observation_model))
    self.observation_model = observation_model
    self.width = int(width)
    self.depth = int(depth)
    if self.depth < 1:
      raise ValueError('depth must be >= 1')
    self.input_scales = np.asarray(input_scales, dtype=np.float64).reshape(-1)
    self.D = self.input_scales.size
    self.fourier_degrees = np.asarray(fourier_degrees, dtype=np.int32).reshape(-1)
    if self.fourier_degrees.size != self.D:
      raise ValueError('fourier_degrees must have one entry per input column')
    self.interactions = np.asarray(interactions, dtype=np.int32).reshape(-1, 2)
    if self.interactions.size and (self.interactions.min() < 0 or
                                   self.interactions.max() >= self.D):
      raise ValueError('interaction index out of range')
    self.freqs, self.harmonics = seasonal_frequency_table(
        np.asarray(seasonality_periods, dtype=float),
        np.asarray(num_seasonal_harmonics))

    # ---- feature columns (models.py:242-252) ----------------------------
    unfiltered = [(GROUP_INPUT, -1, self.D)]
    unfiltered += [(GROUP_FOURIER, d, 2 * int(k))
                   for d, k in enumerate(self.fourier_degrees) if k > 0]
    unfiltered.append((GROUP_SEASONAL, -1, 2 * self.freqs.size))
    unfiltered.append((GROUP_INTERACT, -1, self.interactions.shape[0]))
    kept = [(i, g) for i, g in enumerate(unfiltered) if g[2] > 0]
    self.F = sum(g[2] for _, g in kept)

    # ---- leaves, string-sorted like a flax dict -------------------------
    W = self.width
    tree = {}
    fan_in = self.F
    for l in range(self.depth):
      tree[f'Dense_{l}'] = [('bias', (W,)), ('kernel', (fan_in, W))]
      tree[f'inv_sp_layer_scale{l}'] = ()
      fan_in = W
    tree[f'Dense_{self.depth}'] = [('bias', (1,)), ('kernel', (W, 1))]
    for i, _ in kept:
      tree[f'feature_inv_sp_scale{i}'] = ()
    tree['inv_sp_output_scale'] = ()
    tree['log_scale_adjustment'] = (self.D,)
    tree['logit_activation_weight'] = ()

    flat = [('log_noise_scale', ()), ('shape', ()), ('inflated_loc_probs', ())]
    for key in sorted(tree):
      if isinstance(tree[key], list):
        flat += [(f'{key}/{sub}', shp) for sub, shp in tree[key]]
      else:
        flat.append((key, tree[key]))
    self.leaves = []
    cursor = 0
    for name, shp in flat:
      n = int(np.prod(shp, dtype=np.int64)) if len(shp) else 1
      self.leaves.append(LeafSpec(name, tuple(shp), cursor, n))
      cursor += n
    self.P = cursor
    self.by_name = {lf.name: lf for lf in self.leaves}

    self.groups = []
    col = 0
    for i, (kind, arg, ncols) in kept:
      self.groups.append(GroupSpec(
          kind, arg, ncols, col, self.by_name[f'feature_inv_sp_scale{i}'].offset))
      col += ncols

  # ---- helpers used by the host side -------------------------------------
  def offset(self, name):
    return self.by_name[name].offset

  def matrix_mask(self):
    """True where the entry belongs to a rank-2 leaf (Dense kernels): those
    are the ones drawn from TruncatedNormal at init (inference.py:411-423)."""
    m = np.zeros(self.P, dtype=bool)
    for lf in self.leaves:
      if len(lf.shape) == 2:
        m[lf.offset:lf.offset + lf.size] = True
    return m

  def unpack(self, theta):
    """(..., P) array -> list of (..., *shape) arrays in leaf order."""
    theta = np.asarray(theta)
    lead = theta.shape[:-1]
    return [theta[..., lf.offset:lf.offset + lf.size].reshape(lead + lf.shape)
            for lf in self.leaves]

  def pack(self, leaves, dtype=np.float32):
    """Inverse of `unpack`."""
    first = np.asarray(leaves[0])
    lead = first.shape[:first.ndim - len(self.leaves[0].shape)]
    out = np.empty(lead + (self.P,), dtype=dtype)
    for lf, arr in zip(self.leaves, leaves):
      out[..., lf.offset:lf.offset + lf.size] = np.asarray(arr).reshape(
          lead + (lf.size,))
    return out

  def struct_tuple_type(self):
    """namedtuple type with fields var0..var{n-1}, mimicking TFP's StructTuple
    that the reference returns as `params_` (inference.py:452, models.py:95-103)."""
    return collections.namedtuple(
        'StructTuple', [f'var{i}' for i in range(len(self.leaves))])

  def flops_per_member_step(self, batch, samples=1):
    """Algorithmic FLOPs of one train step of one member:
    6*S*B*(F*W + (depth-1)*W^2 + W)."""
    W = self.width
    return 6.0 * samples * batch * (self.F * W + (self.depth - 1) * W * W + W)

  def min_bytes_per_member_step(self, batch, vi=False):
    """Minimum HBM bytes of one member step: 24*P_opt + 4*B*(D+1)."""
    p_opt = self.P * (2 if vi else 1)
    return 24.0 * p_opt + 4.0 * batch * (self.D + 1)
