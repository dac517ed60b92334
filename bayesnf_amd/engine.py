"""Thin Python owner of one `bnf_handle` and of the device buffers it uses.

torch-ROCm supplies device memory and the HIP stream; every computation is a
call into libbnf_hip.so through ctypes (`_native`).  No computation here.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _native
from . import distributed
from .spec import NetSpec


_warned_default_dtype = False


def default_dtype(compute_dtype=None) -> str:
  """Canonical name of the engine arithmetic.
  'fp32' (= 'f32', 'float32', 'fp32_exact'; include/bnf.h BNF_DTYPE_F32): f32 storage, accumulation and epilogues,
  contractions on the exact `v_mfma_f32_32x32x2_f32` -- an explicit 'fp32' always means this.
  'fp32_split' (= 'bf16x3'; BNF_DTYPE_F32S): the same storage and epilogues, every contraction as three bf16 MFMAs on
  operands split in registers (16 operand bits, ~5e-6 per contraction where the exact chain measures 1e-7), 1.8x the exact
  chain.  WHAT THE ESTIMATORS RUN WHEN NOTHING IS SAID (compute_dtype=None and no BNF_DTYPE): both f32-class engines hold
  SURVEY 8d's fp32 gates verbatim -- forward / loss 1e-5, gradients 1e-4, parameters after 100 full-batch Adam steps 1e-3 on
  the C2 and C4 layouts (tests/util.py FP32_GATE; measured profiles/r06_fp32_contract_diag.txt: split 3.1e-6 / 2.4e-7 /
  7.7e-5 / 2.7e-5) -- and both reproduce the reference's golden predictions to < 1e-4, so the default is the faster one
  (the condition VERDICT r05 item 2 set for keeping it); the once-per-process warning below says so.
  'bf16': the throughput path (bf16 contraction operands, f32 accumulation -- the numerics class of the reference's TPU
  runs; ~6x the exact chain at the benchmark size).  'fp8' = 'bf16' with FP8 OPERAND STORAGE for the weight-gradient
  contractions and, on the folded two-layer row-panel forms, the W x W forward / backward-data contractions on the fp8 MFMA
  (include/bnf.h BNF_DTYPE_FP8; training handles on the row-panel pipeline only -- a forward-only handle of an 'fp8'
  estimator runs the bf16 forward)."""
  global _warned_default_dtype
  if compute_dtype is None and 'BNF_DTYPE' not in os.environ and not _warned_default_dtype:
    _warned_default_dtype = True
    import warnings
    warnings.warn("bayesnf_amd: compute_dtype defaults to 'fp32_split' (f32 storage and epilogues, contractions as three "
                  "split-bf16 MFMAs: within SURVEY 8d's fp32 gates, 1.8x the exact chain). Pass compute_dtype='fp32' for the exact "
                  "f32 MFMA chain, 'bf16' (~3.5x faster again) or 'fp8' -- or set BNF_DTYPE.", stacklevel=3)
  dt = compute_dtype or os.environ.get('BNF_DTYPE', 'fp32_split')
  if dt not in _native.DTYPE:
    raise ValueError(f'compute_dtype must be one of {sorted(_native.DTYPE)}')
  return _native.DTYPE_NAME[_native.DTYPE[dt]]


def _ptr(t):
  return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
  """One GPU's share of an ensemble: `members` networks trained side by side.

  mode 'map' (MAP / MLE via prior_weight) or 'vi'.  X (N, D) and y (N,) are
  copied to the device once, as float32 (what `jnp.array` does with x64 off,
  reference inference.py:553-554).
  """

  def __init__(self, net: NetSpec, *, mode='map', X=None, y=None, batch=None,
               members=1, member_offset=0, seed=0, learning_rate=0.005,
               prior_weight=1.0, kl_weight=1.0, vi_samples=1,
               compute_dtype=None, forward_only=False, row_capacity=None,
               device_index=None, pipeline='auto'):
    self.lib = _native.load()
    if not torch.cuda.is_available():
      raise RuntimeError(
          'bayesnf_amd: no GPU visible (torch.cuda.is_available() is False). '
          'The engine is HIP-only; there is no CPU fallback.')
    self.net = net
    self.mode = mode
    self.dtype = default_dtype(compute_dtype)
    # default device: the first one this process drives (LOCAL_RANK's under torch.distributed, else the first of
    # BNF_DEVICES / of the visible devices) -- the same list fit() / predict() deal members over
    self.dev_index = (distributed.local_devices()[0]
                      if device_index is None else int(device_index))
    self.device = torch.device(f'cuda:{self.dev_index}')
    torch.cuda.set_device(self.device)
    self.members = int(members)
    self.member_offset = int(member_offset)
    self.forward_only = bool(forward_only)
    if forward_only:
      n_rows = batch = int(row_capacity)
      self.X = self.y = None
    else:
      X = np.ascontiguousarray(np.asarray(X, dtype=np.float64), dtype=np.float32)
      y = np.ascontiguousarray(np.asarray(y, dtype=np.float64), dtype=np.float32)
      n_rows = X.shape[0]
      batch = n_rows if batch is None else int(batch)
      self.X = torch.from_numpy(X).to(self.device)
      self.y = torch.from_numpy(y).to(self.device)
    self.n_rows, self.batch = n_rows, batch
    self.S = int(vi_samples) if mode == 'vi' else 1
    cfg = _native.make_config(
        net, device=self.dev_index, dtype=self.dtype,
        mode=_native.MODE_VI if mode == 'vi' else _native.MODE_MAP,
        n_rows=n_rows, batch=batch, members=members,
        member_offset=member_offset, seed=_native.seed_to_u64(seed),
        learning_rate=learning_rate, prior_weight=prior_weight,
        kl_weight=kl_weight, vi_samples=self.S, forward_only=forward_only,
        pipeline=pipeline)
    self.cfg = cfg
    handle = C.c_void_p()
    _native.check(self.lib.bnf_create(C.byref(cfg), C.byref(handle)), 'bnf_create')
    self.handle = handle
    ws_bytes = self.lib.bnf_workspace_bytes(handle)
    self.workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
    if forward_only:
      self.params = self.state = None
    else:
      self.params = torch.empty(self.lib.bnf_param_bytes(handle) // 4,
                                dtype=torch.float32, device=self.device)
      self.state = torch.empty(self.lib.bnf_state_bytes(handle) // 4,
                               dtype=torch.float32, device=self.device)
    self.stream = torch.cuda.current_stream(self.device)
    _native.check(
        self.lib.bnf_bind(handle, _ptr(self.params), _ptr(self.state),
                          _ptr(self.workspace), _ptr(self.X), _ptr(self.y),
                          C.c_void_p(self.stream.cuda_stream)), 'bnf_bind')

  def owned_bytes(self) -> int:
    """Device memory the engine allocated itself (include/bnf.h bnf_owned_bytes): the row-key work buffers."""
    return int(self.lib.bnf_owned_bytes(self.handle))

  # -- lifecycle ---------------------------------------------------------------
  def close(self):
    if getattr(self, 'handle', None) is not None:
      torch.cuda.synchronize(self.device)
      self.lib.bnf_destroy(self.handle)
      self.handle = None
      self.workspace = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  # -- training ----------------------------------------------------------------
  def init_params(self, log_noise_init: float):
    _native.check(self.lib.bnf_init_params(self.handle, C.c_float(log_noise_init)),
                  'bnf_init_params')

  def init_params_keys(self, leaf_keys, log_noise_init: float = 0.0):
    """The reference's own initial parameters (VI: surrogate means) drawn on the device from the members' per-leaf
    keys (include/bnf.h bnf_init_params_keys): uint32 (members, n_leaves, 2) from jaxseed.map_leaf_keys /
    vi_mean_leaf_keys."""
    from . import jaxseed
    k = np.ascontiguousarray(leaf_keys, dtype=np.uint32)
    n_leaves = len(self.net.leaves)
    if k.shape != (self.members, n_leaves, 2):
      raise ValueError(f'leaf keys must be uint32 ({self.members}, {n_leaves}, 2); got {k.shape}')
    t = torch.from_numpy(k.view(np.int32)).to(self.device)
    off = jaxseed.leaf_offsets(self.net)
    arr = (C.c_int32 * len(off))(*[int(v) for v in off])
    _native.check(self.lib.bnf_init_params_keys(self.handle, _ptr(t), arr, n_leaves, float(log_noise_init)),
                  'bnf_init_params_keys')
    torch.cuda.synchronize(self.device)     # `t` is released on return

  def set_params(self, theta):
    """theta: array (members, P) [MAP] or (2, members, P) = (mu, rho) [VI]."""
    t = torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float32))
    self.params.copy_(t.reshape(-1).to(self.device))

  def get_params(self) -> np.ndarray:
    p = self.params.detach().cpu().numpy()
    if self.mode == 'vi':
      return p.reshape(2, self.members, self.net.P)
    return p.reshape(self.members, self.net.P)

  def train(self, epoch0: int, num_epochs: int) -> torch.Tensor:
    """Enqueue the whole optimisation; returns the (members, num_epochs) loss
    tensor on the device (valid after a stream sync)."""
    losses = torch.zeros((self.members, num_epochs), dtype=torch.float32,
                         device=self.device)
    if num_epochs > 0:
      _native.check(self.lib.bnf_train(self.handle, int(epoch0), int(num_epochs),
                                       _ptr(losses)), 'bnf_train')
    return losses

  def set_row_tables(self, tables, epoch0=0):
    """Epoch shuffles supplied by the caller (include/bnf.h bnf_row_tables): int32 (n_epochs, members,
    (N // batch) * batch) row ids for the epochs [epoch0, epoch0 + n_epochs); uploaded and kept alive
    here (together with the previous chunk, which may still be executing).  None restores the engine's
    own shuffle."""
    if tables is None:
      self._row_tables = []
      _native.check(self.lib.bnf_row_tables(self.handle, None, 0, 0), 'bnf_row_tables')
      return
    t = tables if isinstance(tables, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(tables, dtype=np.int32))
    t = t.to(self.device, non_blocking=True).contiguous()
    keep = (self.n_rows // self.batch) * self.batch
    if t.dtype != torch.int32 or t.dim() != 3 or t.shape[1] != self.members or t.shape[2] != keep:
      raise ValueError(f'row tables must be int32 (n_epochs, {self.members}, {keep}); got {tuple(t.shape)} {t.dtype}')
    self._row_tables = (getattr(self, '_row_tables', []) + [t])[-2:]
    _native.check(self.lib.bnf_row_tables(self.handle, _ptr(t), int(epoch0), int(t.shape[0])), 'bnf_row_tables')

  def set_row_keys(self, subkeys, epoch0=0):
    """The reference's epoch shuffles drawn on the device (include/bnf.h bnf_row_keys): uint32 (n_epochs, members,
    rounds, 2) sub keys of `jax.random.permutation`'s sort rounds (jaxseed.map_shuffle_subkeys) for the epochs
    [epoch0, epoch0 + n_epochs); uploaded and kept alive here.  None switches it off."""
    if subkeys is None:
      self._row_keys = None
      _native.check(self.lib.bnf_row_keys(self.handle, None, 0, 0, 0), 'bnf_row_keys')
      return
    k = np.ascontiguousarray(subkeys, dtype=np.uint32)
    per = 1 if self.mode == 'vi' else self.members     # VI: one batch per step, shared by every member (jaxseed.vi_batch_subkeys)
    if k.ndim != 4 or k.shape[1] != per or k.shape[3] != 2:
      raise ValueError(f'row keys must be uint32 (n_epochs, {per}, rounds, 2); got {k.shape}')
    t = torch.from_numpy(k.view(np.int32)).to(self.device)
    self._row_keys = t
    _native.check(self.lib.bnf_row_keys(self.handle, _ptr(t), int(epoch0), int(k.shape[0]), int(k.shape[2])), 'bnf_row_keys')

  def vi_posterior_draws(self, n_draws: int) -> torch.Tensor:
    out = torch.empty((n_draws, self.members, self.net.P), dtype=torch.float32,
                      device=self.device)
    _native.check(self.lib.bnf_vi_posterior_draws(self.handle, int(n_draws), _ptr(out)),
                  'bnf_vi_posterior_draws')
    return out

  # -- prediction --------------------------------------------------------------
  def forward(self, theta: torch.Tensor, Xnew: torch.Tensor):
    """theta (M, P) f32 device, Xnew (R, D) f32 device -> loc (M, R), aux (M, 3)."""
    theta = theta.contiguous()
    Xnew = Xnew.contiguous()
    M, R = theta.shape[0], Xnew.shape[0]
    loc = torch.empty((M, R), dtype=torch.float32, device=self.device)
    aux = torch.empty((M, 3), dtype=torch.float32, device=self.device)
    _native.check(self.lib.bnf_forward(self.handle, _ptr(theta), M, _ptr(Xnew), R,
                                       _ptr(loc), _ptr(aux)), 'bnf_forward')
    return loc, aux

  def normal_mixture_quantiles(self, means: torch.Tensor, scales: torch.Tensor,
                               quantiles, approximate=False) -> torch.Tensor:
    """means (M, R), scales (M,) -> (n_q, R)."""
    means = means.contiguous().float()
    scales = scales.contiguous().float()
    q = np.asarray(list(quantiles), dtype=np.float32)
    out = torch.empty((len(q), means.shape[1]), dtype=torch.float32, device=self.device)
    qa = (C.c_float * len(q))(*q.tolist())
    _native.check(self.lib.bnf_normal_mixture_quantiles(
        self.handle, _ptr(means), _ptr(scales), means.shape[0], means.shape[1], qa,
        len(q), 1 if approximate else 0, _ptr(out)), 'bnf_normal_mixture_quantiles')
    return out

  def count_mixture_quantiles(self, loc: torch.Tensor, aux: torch.Tensor, quantiles):
    """NB / ZINB: loc (M, R) network output, aux (M, 3) -> (means (M, R), quantiles (n_q, R))."""
    loc = loc.contiguous().float()
    aux = aux.contiguous().float()
    q = np.asarray(list(quantiles), dtype=np.float32)
    means = torch.empty_like(loc)
    out = torch.empty((len(q), loc.shape[1]), dtype=torch.float32, device=self.device)
    qa = (C.c_float * max(1, len(q)))(*q.tolist())
    _native.check(self.lib.bnf_count_mixture_quantiles(
        self.handle, _ptr(loc), _ptr(aux), loc.shape[0], loc.shape[1], qa, len(q),
        _ptr(means), _ptr(out)), 'bnf_count_mixture_quantiles')
    return means, out

  # -- introspection (tests, bench) -------------------------------------------
  def debug_loss_and_grad(self, epoch=0, step=0):
    k = 2 if self.mode == 'vi' else 1
    grads = torch.empty((k * self.members, self.net.P), dtype=torch.float32, device=self.device)
    loss = torch.empty((self.members,), dtype=torch.float32, device=self.device)
    _native.check(self.lib.bnf_debug_loss_and_grad(self.handle, epoch, step, _ptr(grads),
                                                   _ptr(loss)), 'bnf_debug_loss_and_grad')
    torch.cuda.synchronize(self.device)
    g = grads.cpu().numpy()
    if self.mode == 'vi':
      g = g.reshape(2, self.members, self.net.P)
    return loss.cpu().numpy(), g

  def debug_row_index(self, epoch=0, step=0) -> np.ndarray:
    out = torch.empty((self.members, self.batch), dtype=torch.int32, device=self.device)
    _native.check(self.lib.bnf_debug_row_index(self.handle, epoch, step, _ptr(out)),
                  'bnf_debug_row_index')
    torch.cuda.synchronize(self.device)
    return out.cpu().numpy()

  def debug_vi_eps(self, step=0) -> np.ndarray:
    out = torch.empty((self.members, self.S, self.net.P), dtype=torch.float32, device=self.device)
    _native.check(self.lib.bnf_debug_vi_eps(self.handle, step, _ptr(out)), 'bnf_debug_vi_eps')
    torch.cuda.synchronize(self.device)
    return out.cpu().numpy()

  def set_vi_noise_keys(self, step_keys, draw_keys, leaf_offsets):
    """The reference's VI noise stream (include/bnf.h bnf_vi_noise_keys): key tables from
    jaxseed.vi_noise_keys / vi_draw_keys (uint32 numpy), uploaded and kept alive here."""
    if step_keys is None and draw_keys is None:
      self._vi_keys = None
      _native.check(self.lib.bnf_vi_noise_keys(self.handle, None, 0, None, 0, None, 0), 'bnf_vi_noise_keys')
      return
    up = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32)).to(self.device)
    sk, dk = up(step_keys), up(draw_keys)
    lo = np.ascontiguousarray(leaf_offsets, dtype=np.int32)
    self._vi_keys = (sk, dk)
    _native.check(self.lib.bnf_vi_noise_keys(
        self.handle, None if sk is None else _ptr(sk), 0 if sk is None else int(step_keys.shape[0]),
        None if dk is None else _ptr(dk), 0 if dk is None else int(draw_keys.shape[0]),
        lo.ctypes.data_as(C.c_void_p), int(len(lo) - 1)), 'bnf_vi_noise_keys')

  def debug_vi_noise(self, eps):
    """Verification hook (include/bnf.h bnf_debug_vi_noise): every following VI step reads its
    (members, S, P) standard normals from `eps` (device tensor, kept alive here); None restores
    the engine's generator."""
    self._ext_eps = None if eps is None else eps.contiguous().float().to(self.device)
    _native.check(self.lib.bnf_debug_vi_noise(self.handle, None if eps is None else _ptr(self._ext_eps)),
                  'bnf_debug_vi_noise')

  def debug_poison_lds(self, pattern=0x7fc00000):
    """Test hook (include/bnf.h bnf_debug_poison_lds): garbage in every CU's LDS before the next kernels."""
    _native.check(self.lib.bnf_debug_poison_lds(self.handle, int(pattern) & 0xffffffff), 'bnf_debug_poison_lds')

  def debug_activation(self, what: int) -> np.ndarray:
    ev = self.members * self.S
    if what == 0 or what == 400:
      shape = (ev, self.batch, self.net.F)
    elif what == 200:
      shape = (ev, self.batch)
    else:
      shape = (ev, self.batch, self.net.width)
    out = torch.empty(shape, dtype=torch.float32, device=self.device)
    _native.check(self.lib.bnf_debug_activation(self.handle, what, _ptr(out)),
                  'bnf_debug_activation')
    torch.cuda.synchronize(self.device)
    return out.cpu().numpy()

  def debug_gemm_nt(self, A: np.ndarray, Bt: np.ndarray) -> np.ndarray:
    A = torch.as_tensor(np.ascontiguousarray(A, dtype=np.float32)).to(self.device)
    Bt = torch.as_tensor(np.ascontiguousarray(Bt, dtype=np.float32)).to(self.device)
    Cout = torch.empty((A.shape[0], Bt.shape[0]), dtype=torch.float32, device=self.device)
    _native.check(self.lib.bnf_debug_gemm_nt(self.handle, _ptr(A), _ptr(Bt), A.shape[0],
                                             Bt.shape[0], A.shape[1], _ptr(Cout)),
                  'bnf_debug_gemm_nt')
    torch.cuda.synchronize(self.device)
    return Cout.cpu().numpy()

  def debug_gemm_tn(self, A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """A (R, M), B (R, N) row-major -> A^T B (M, N) through the weight-gradient core."""
    A = torch.as_tensor(np.ascontiguousarray(A, dtype=np.float32)).to(self.device)
    B = torch.as_tensor(np.ascontiguousarray(B, dtype=np.float32)).to(self.device)
    Cout = torch.empty((A.shape[1], B.shape[1]), dtype=torch.float32, device=self.device)
    _native.check(self.lib.bnf_debug_gemm_tn(self.handle, _ptr(A), _ptr(B), A.shape[0], A.shape[1],
                                             B.shape[1], _ptr(Cout)), 'bnf_debug_gemm_tn')
    torch.cuda.synchronize(self.device)
    return Cout.cpu().numpy()

  def profile(self, kernel):
    """kernel: '*' (all), a kernel name, or None (off)."""
    arg = None if kernel is None else str(kernel).encode()
    _native.check(self.lib.bnf_profile_enable(self.handle, arg), 'bnf_profile_enable')

  def profile_read(self) -> dict:
    cap = 32
    n = C.c_int32(cap)
    names = (C.c_char_p * cap)()
    avg = (C.c_double * cap)()
    calls = (C.c_int64 * cap)()
    _native.check(self.lib.bnf_profile_read(self.handle, C.byref(n), names, avg, calls),
                  'bnf_profile_read')
    out = {}
    for i in range(n.value):
      name = names[i].decode()
      out[name] = dict(avg_ms=avg[i], calls=calls[i],
                       flops=self.lib.bnf_kernel_flops(self.handle, names[i]))
    return out
