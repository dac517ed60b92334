"""A CPU stand-in for `bayesnf_amd.engine.Engine`, for the tests of the N > 1 plumbing that must run without a GPU.

TEST INFRASTRUCTURE ONLY (the product has no CPU path: bayesnf_amd.engine.Engine refuses without the HIP library).
It is injected through ONE seam -- the name `Engine` in `bayesnf_amd.inference` -- and does real work behind the interface
`fit_map` uses, so that a sharded fit flows through run_shards -> gather_shards -> _struct_tuple with numbers that mean
something: initial parameters from the per-leaf keys `fit_map` hands it (the reference's chain, jaxseed), full-batch or
minibatch MAP / MLE training by the float64 ORACLE (oracle/bnf_oracle.py train_map), minibatch rows replayed from the
sort-round sub keys `fit_map` hands `set_row_keys`.  Every result is a pure function of the GLOBAL member id."""
import numpy as np
import torch

from bayesnf_amd import jaxseed
from oracle import bnf_oracle as O
from oracle import jax_rng as R


class StandInEngine:
  created = []          # (member_offset, members, device_index) of every instance, for the tests
  model_kwargs = {}     # NetSpec / oracle Model keyword arguments of the network under test

  def __init__(self, net, *, mode='map', X=None, y=None, batch=None, members=1, member_offset=0, seed=0,
               learning_rate=0.005, prior_weight=1.0, compute_dtype=None, device_index=None, **_):
    assert mode == 'map'
    self.net, self.members, self.member_offset = net, int(members), int(member_offset)
    self.X, self.y = np.asarray(X, dtype=np.float64), np.asarray(y, dtype=np.float64)
    self.n_rows = self.y.shape[0]
    self.batch = self.n_rows if batch is None else int(batch)
    self.lr, self.pw = float(learning_rate), float(prior_weight)
    self.model = O.Model(**StandInEngine.model_kwargs)       # the oracle's description of the same network (set by the test)
    assert self.model.P == net.P
    self.theta = np.zeros((self.members, net.P))
    self._sub = None
    StandInEngine.created.append((self.member_offset, self.members, device_index))

  def init_params_keys(self, leaf_keys, log_noise_init):
    th = np.zeros((self.members, self.net.P), dtype=np.float32)
    th[:, self.net.by_name['log_noise_scale'].offset] = np.float32(log_noise_init)
    for e in range(self.members):
      for i, lf in enumerate(self.net.leaves):
        if len(lf.shape) == 2:
          th[e, lf.offset:lf.offset + lf.size] = jaxseed.truncated_normal_std(leaf_keys[e, i], lf.size)
    self.theta = th.astype(np.float64)

  def set_row_keys(self, subkeys, epoch0=0):
    assert epoch0 == 0
    self._sub = np.asarray(subkeys, dtype=np.uint32)          # (n_epochs, members, rounds, 2)

  def _rows(self, ep):
    out = np.empty((self.members, self.n_rows), dtype=np.int64)
    for e in range(self.members):
      x = np.arange(self.n_rows)
      for r in range(self._sub.shape[2]):
        x = x[np.argsort(R.random_bits(self._sub[ep, e, r], (self.n_rows,)), kind='stable')]
      out[e] = x
    return out[:, :(self.n_rows // self.batch) * self.batch]

  def train(self, epoch0, n_epochs):
    assert epoch0 == 0
    kw = dict(batch_size=self.batch, row_index_fn=self._rows) if self.batch < self.n_rows else {}
    self.theta, losses = O.train_map(self.model, self.theta, self.X, self.y, lr=self.lr, num_epochs=n_epochs,
                                     prior_weight=self.pw, **kw)
    return torch.from_numpy(np.asarray(losses, dtype=np.float32))

  @property
  def params(self):
    return torch.from_numpy(np.ascontiguousarray(self.theta, dtype=np.float32)).reshape(-1)

  def close(self):
    pass
