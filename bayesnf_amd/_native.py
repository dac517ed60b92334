"""ctypes binding of libbnf_hip.so (C ABI: include/bnf.h).

There is deliberately no fallback: if the shared library (or a gfx950 device)
is missing every compute call raises.  torch-ROCm tensors are used only as
device-memory containers; the library receives raw pointers.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import spec as _spec

_LIB_NAME = 'libbnf_hip.so'
_lib = None

ABI_VERSION = 6
MAX_INPUTS, MAX_GROUPS, MAX_LAYERS, MAX_FREQS, MAX_INTERACT = 8, 12, 8, 96, 16
# 'fp32' (= 'f32', 'float32', 'fp32_exact') = BNF_DTYPE_F32: the exact f32 MFMA chain -- what an explicit 'fp32' always means
# (round 5 had silently mapped it to the split form: ADVICE r05); 'fp32_split' (= 'bf16x3') = BNF_DTYPE_F32S: f32 storage /
# accumulation / epilogues with the contractions on split-bf16 MFMAs (16 operand bits, ~5e-6 per contraction against 1e-7;
# 1.8x the exact chain) -- selected by name, or by saying nothing (engine.default_dtype: the estimators' default, announced
# in its warning; both hold SURVEY 8d's fp32 gates verbatim)
DTYPE = {'fp32': 0, 'f32': 0, 'float32': 0, 'fp32_exact': 0, 'fp32_split': 3, 'bf16x3': 3, 'bf16': 1, 'bfloat16': 1, 'fp8': 2}
DTYPE_NAME = {0: 'fp32', 1: 'bf16', 2: 'fp8', 3: 'fp32_split'}
OBS = {'NORMAL': 0, 'NB': 1, 'ZINB': 2}
MODE_MAP, MODE_VI = 0, 1
PIPELINE = {'auto': 0, 'layers': 1, 'panel': 3}

EXPORTS = (
    'bnf_abi_version', 'bnf_last_error', 'bnf_create', 'bnf_destroy',
    'bnf_workspace_bytes', 'bnf_state_bytes', 'bnf_param_bytes', 'bnf_owned_bytes', 'bnf_bind',
    'bnf_init_params', 'bnf_init_params_keys', 'bnf_train', 'bnf_row_tables', 'bnf_row_keys', 'bnf_vi_posterior_draws', 'bnf_vi_noise_keys', 'bnf_forward',
    'bnf_normal_mixture_quantiles', 'bnf_count_mixture_quantiles', 'bnf_debug_loss_and_grad',
    'bnf_debug_row_index', 'bnf_debug_vi_eps', 'bnf_debug_vi_noise', 'bnf_debug_activation',
    'bnf_debug_gemm_nt', 'bnf_debug_gemm_tn', 'bnf_debug_poison_lds', 'bnf_profile_enable', 'bnf_profile_read',
    'bnf_kernel_flops', 'bnf_comm_available', 'bnf_comm_unique_id', 'bnf_comm_create', 'bnf_allgather',
    'bnf_comm_create_local', 'bnf_allgather_group', 'bnf_comm_destroy')


class BnfConfig(C.Structure):
  """Mirror of `struct bnf_config` (include/bnf.h); field order matters."""
  _fields_ = [
      ('abi_version', C.c_int32), ('device', C.c_int32), ('dtype', C.c_int32),
      ('obs_model', C.c_int32), ('mode', C.c_int32),
      ('n_inputs', C.c_int32), ('width', C.c_int32), ('depth', C.c_int32),
      ('n_features', C.c_int32), ('n_params', C.c_int32),
      ('n_groups', C.c_int32),
      ('group_kind', C.c_int32 * MAX_GROUPS),
      ('group_arg', C.c_int32 * MAX_GROUPS),
      ('group_ncols', C.c_int32 * MAX_GROUPS),
      ('group_col0', C.c_int32 * MAX_GROUPS),
      ('group_scale_off', C.c_int32 * MAX_GROUPS),
      ('fourier_degree', C.c_int32 * MAX_INPUTS),
      ('input_scale', C.c_float * MAX_INPUTS),
      ('n_freqs', C.c_int32),
      ('freq', C.c_float * MAX_FREQS),
      ('harmonic', C.c_float * MAX_FREQS),
      ('n_interact', C.c_int32),
      ('interact', (C.c_int32 * 2) * MAX_INTERACT),
      ('off_log_noise_scale', C.c_int32), ('off_shape', C.c_int32),
      ('off_inflated', C.c_int32),
      ('off_bias', C.c_int32 * (MAX_LAYERS + 1)),
      ('off_kernel', C.c_int32 * (MAX_LAYERS + 1)),
      ('off_layer_scale', C.c_int32 * MAX_LAYERS),
      ('off_output_scale', C.c_int32), ('off_lsa', C.c_int32),
      ('off_act_weight', C.c_int32),
      ('n_rows', C.c_int64), ('batch', C.c_int64),
      ('members', C.c_int32), ('member_offset', C.c_int64),
      ('vi_samples', C.c_int32), ('forward_only', C.c_int32), ('pipeline', C.c_int32),
      ('learning_rate', C.c_float), ('prior_weight', C.c_float),
      ('kl_weight', C.c_float),
      ('seed', C.c_uint64),
  ]


def library_path() -> str:
  """In-tree shared object; BNF_LIB=<path> selects another build (A/B perf experiments)."""
  override = os.environ.get('BNF_LIB')
  if override:
    return override
  return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load():
  """dlopen the engine (once) and declare prototypes.  Raises RuntimeError
  with build instructions when the shared object is missing."""
  global _lib
  if _lib is not None:
    return _lib
  path = library_path()
  if not os.path.exists(path):
    raise RuntimeError(
        f'{path} not found. Build it with `python -c "import __graft_entry__ '
        'as g; g.build()"` or `make -C bayesnf_amd/csrc`. bayesnf_amd has no '
        'CPU fallback.')
  lib = C.CDLL(path)
  vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p
  lib.bnf_abi_version.restype = C.c_int
  lib.bnf_last_error.restype = C.c_char_p
  lib.bnf_create.argtypes = [C.POINTER(BnfConfig), C.POINTER(vp)]
  lib.bnf_destroy.argtypes = [vp]
  lib.bnf_destroy.restype = None
  for name in ('bnf_workspace_bytes', 'bnf_state_bytes', 'bnf_param_bytes', 'bnf_owned_bytes'):
    getattr(lib, name).argtypes = [vp]
    getattr(lib, name).restype = C.c_size_t
  lib.bnf_bind.argtypes = [vp, vp, vp, vp, vp, vp, vp]
  lib.bnf_init_params.argtypes = [vp, C.c_float]
  lib.bnf_train.argtypes = [vp, i64, i64, f32p]
  lib.bnf_comm_unique_id.argtypes = [vp]
  lib.bnf_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
  lib.bnf_allgather.argtypes = [vp, vp, vp, C.c_size_t, vp]
  lib.bnf_comm_available.argtypes = []
  lib.bnf_comm_create_local.argtypes = [i32, C.POINTER(C.c_int32), C.POINTER(vp)]
  lib.bnf_allgather_group.argtypes = [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_size_t, C.POINTER(vp)]
  lib.bnf_comm_destroy.argtypes = [vp]
  lib.bnf_comm_destroy.restype = None
  lib.bnf_vi_posterior_draws.argtypes = [vp, i32, f32p]
  lib.bnf_forward.argtypes = [vp, vp, i64, vp, i64, vp, vp]
  lib.bnf_normal_mixture_quantiles.argtypes = [
      vp, vp, vp, i64, i64, C.POINTER(C.c_float), i32, i32, vp]
  lib.bnf_count_mixture_quantiles.argtypes = [
      vp, vp, vp, i64, i64, C.POINTER(C.c_float), i32, vp, vp]
  lib.bnf_debug_loss_and_grad.argtypes = [vp, i64, i64, vp, vp]
  lib.bnf_debug_row_index.argtypes = [vp, i64, i64, vp]
  lib.bnf_debug_vi_eps.argtypes = [vp, i64, vp]
  lib.bnf_debug_vi_noise.argtypes = [vp, vp]
  lib.bnf_debug_poison_lds.argtypes = [vp, C.c_uint32]
  lib.bnf_row_tables.argtypes = [vp, vp, i64, i64]
  lib.bnf_row_keys.argtypes = [vp, vp, i64, i64, C.c_int32]
  lib.bnf_init_params_keys.argtypes = [vp, vp, C.POINTER(C.c_int32), C.c_int32, C.c_float]
  lib.bnf_vi_noise_keys.argtypes = [vp, vp, i64, vp, i64, vp, C.c_int32]
  lib.bnf_debug_activation.argtypes = [vp, i32, vp]
  lib.bnf_debug_gemm_nt.argtypes = [vp, vp, vp, i32, i32, i32, vp]
  lib.bnf_debug_gemm_tn.argtypes = [vp, vp, vp, i32, i32, i32, vp]
  lib.bnf_profile_enable.argtypes = [vp, C.c_char_p]
  lib.bnf_profile_read.argtypes = [
      vp, C.POINTER(i32), C.POINTER(C.c_char_p), C.POINTER(C.c_double),
      C.POINTER(i64)]
  lib.bnf_kernel_flops.argtypes = [vp, C.c_char_p]
  lib.bnf_kernel_flops.restype = C.c_double
  for name in EXPORTS:
    getattr(lib, name)  # AttributeError here = stale .so missing an ABI symbol
  if lib.bnf_abi_version() != ABI_VERSION:
    raise RuntimeError('libbnf_hip.so ABI version mismatch')
  _lib = lib
  return lib


def last_error() -> str:
  return load().bnf_last_error().decode('utf-8', 'replace')


def check(rc: int, what: str):
  if rc == 0:
    return
  msg = f'{what}: {last_error()} (code {rc})'
  if rc == -1:
    raise ValueError(msg)
  raise RuntimeError(msg)


_comm_cache = {}
_local_comm_cache = {}


def _agree(ok: bool, world: int, device, on_dev: bool) -> bool:
  """True iff `ok` on EVERY rank (one all_reduce MIN): whatever follows is entered by all ranks or by none."""
  import torch
  if world <= 1:
    return ok
  flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
  flag = flag.to(device) if on_dev else flag
  torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
  return int(flag.item()) == 1


def allgather(send, recv, world: int, rank: int, lib=None):
  """RCCL all-gather through the engine library's own entry point (include/bnf.h:
  bnf_allgather): `send` (…) and `recv` (world, …) are contiguous device tensors.  The
  communicator is created once per (world, rank, device): rank 0 makes the id, the existing
  torch.distributed process group only carries those 128 bytes to the other ranks.
  Raises RuntimeError on EVERY rank when any rank cannot take part (the ranks agree before and
  after the communicator is made), so a caller may fall back to another collective without the
  ranks ending up in different ones.
  (`lib`: the loaded library; bench.py's CPU self-test passes a stand-in with the same
  entry points to exercise this plumbing without a GPU.)"""
  import torch
  lib = load() if lib is None else lib
  dev = send.device.index if send.is_cuda else -1
  key = (world, rank, dev, id(lib))
  if _comm_cache.get(key) is None:
    on_dev = send.is_cuda and world > 1 and torch.distributed.get_backend() == 'nccl'
    # Every rank finds out LOCALLY whether the library can reach RCCL (symbol resolution: no id, no bootstrap
    # listener on the ranks that will never serve one); only rank 0 makes the id.
    buf = (C.c_char * 128)()
    rc = lib.bnf_comm_available() if hasattr(lib, 'bnf_comm_available') else 0
    if rc == 0 and rank == 0:
      rc = lib.bnf_comm_unique_id(buf)
    err = None if rc == 0 else (last_error() if hasattr(lib, 'bnf_last_error') else f'code {rc}')
    if not _agree(rc == 0, world, send.device, on_dev):
      raise RuntimeError('RCCL is not reachable through libbnf_hip.so on at least one rank' + (f' (here: {err})' if err else ''))
    ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    if world > 1:
      carrier = ident.to(send.device) if on_dev else ident
      torch.distributed.broadcast(carrier, src=0)
      ident = carrier.cpu()
    comm = C.c_void_p()
    raw = (C.c_char * 128).from_buffer_copy(bytes(ident.numpy().tobytes()))
    rc = lib.bnf_comm_create(raw, world, rank, max(dev, 0), C.byref(comm))
    err = None if rc == 0 else (last_error() if hasattr(lib, 'bnf_last_error') else f'code {rc}')
    if rc == 0:
      _comm_cache[key] = comm
    # (a rank whose ncclCommInitRank failed early leaves the others inside theirs until RCCL's own timeout: nothing a
    # caller can do about that; what CAN be guaranteed is that afterwards all ranks take the same branch)
    if not _agree(rc == 0, world, send.device, on_dev):
      if rc == 0:
        lib.bnf_comm_destroy(_comm_cache.pop(key))
      raise RuntimeError('bnf_comm_create failed on at least one rank' + (f' (here: {err})' if err else ''))
  stream = torch.cuda.current_stream(send.device).cuda_stream if send.is_cuda else 0
  check(lib.bnf_allgather(_comm_cache[key], C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                          C.c_size_t(send.numel() * send.element_size()), C.c_void_p(stream)), 'bnf_allgather')


def allgather_local(sends, recvs, lib=None):
  """ONE process, one tensor per local device: every device's block to every device with one grouped RCCL
  all-gather (include/bnf.h bnf_comm_create_local / bnf_allgather_group -- the single-process mode is the
  reference's own shape, inference.py:573-579).  sends[i] (…) on device i of the job, recvs[i] (n, …) on the
  same device.  The communicator set is made once per device list.  Raises RuntimeError when RCCL cannot
  serve the list (e.g. a device named twice: BNF_DEVICES=0,0)."""
  import torch
  lib = load() if lib is None else lib
  n = len(sends)
  devs = tuple((t.device.index if t.is_cuda else i) for i, t in enumerate(sends))
  key = (devs, id(lib))
  if key not in _local_comm_cache:
    arr = (C.c_int32 * n)(*devs)
    comms = (C.c_void_p * n)()
    rc = lib.bnf_comm_create_local(n, arr, comms)
    if rc != 0:
      raise RuntimeError('bnf_comm_create_local: ' + (last_error() if hasattr(lib, 'bnf_last_error') else f'code {rc}'))
    _local_comm_cache[key] = comms
  comms = _local_comm_cache[key]
  nbytes = sends[0].numel() * sends[0].element_size()
  for s_, r_ in zip(sends, recvs):
    if s_.numel() * s_.element_size() != nbytes or r_.numel() * r_.element_size() != n * nbytes:
      raise ValueError('allgather_local: every rank contributes the same byte count')
  sp = (C.c_void_p * n)(*[t.data_ptr() for t in sends])
  rp = (C.c_void_p * n)(*[t.data_ptr() for t in recvs])
  st = (C.c_void_p * n)(*[(torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0) for t in sends])
  rc = lib.bnf_allgather_group(n, comms, sp, rp, C.c_size_t(nbytes), st)
  if rc != 0:
    raise RuntimeError('bnf_allgather_group: ' + (last_error() if hasattr(lib, 'bnf_last_error') else f'code {rc}'))


def seed_to_u64(seed) -> int:
  """int, or anything shaped like a jax PRNGKey (2 x uint32) -> 64-bit seed."""
  if isinstance(seed, (int, np.integer)):
    return int(seed) & 0xFFFFFFFFFFFFFFFF
  arr = np.asarray(seed).astype(np.uint64).reshape(-1)
  if arr.size == 1:
    return int(arr[0])
  if arr.size != 2:
    raise ValueError('seed must be an int or a length-2 uint32 array')
  return (int(arr[0]) << 32) | (int(arr[1]) & 0xFFFFFFFF)


def fold_in(seed_u64: int, data: int) -> int:
  """Derive an independent 64-bit seed (stand-in for jax.random.fold_in used by
  fit_map's num_splits loop, inference.py:434): splitmix64 of seed ^ golden*data."""
  z = (seed_u64 ^ (0x9E3779B97F4A7C15 * (data + 1))) & 0xFFFFFFFFFFFFFFFF
  z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
  z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
  return z ^ (z >> 31)


def make_config(net: _spec.NetSpec, *, device, dtype, mode, n_rows, batch,
                members, member_offset, seed, learning_rate=0.005,
                prior_weight=1.0, kl_weight=1.0, vi_samples=1, forward_only=False,
                pipeline='auto') -> BnfConfig:
  """Serialise a NetSpec + run arguments into the C struct."""
  if net.D > MAX_INPUTS:
    raise ValueError(f'at most {MAX_INPUTS} input columns are supported')
  if len(net.groups) > MAX_GROUPS:
    raise ValueError('too many feature groups')
  if net.depth > MAX_LAYERS:
    raise ValueError(f'depth > {MAX_LAYERS} not supported')
  if net.freqs.size > MAX_FREQS:
    raise ValueError(f'more than {MAX_FREQS} seasonal frequencies')
  if net.interactions.shape[0] > MAX_INTERACT:
    raise ValueError(f'more than {MAX_INTERACT} interactions')
  if not 1 <= net.width <= 8192:
    raise ValueError('width must be in 1..8192 on this backend')
  c = BnfConfig()
  c.abi_version = ABI_VERSION
  c.device = int(device)
  c.dtype = DTYPE[dtype]
  c.obs_model = OBS[net.observation_model]
  c.mode = mode
  c.n_inputs, c.width, c.depth = net.D, net.width, net.depth
  c.n_features, c.n_params, c.n_groups = net.F, net.P, len(net.groups)
  for i, g in enumerate(net.groups):
    c.group_kind[i], c.group_arg[i] = g.kind, g.arg
    c.group_ncols[i], c.group_col0[i] = g.ncols, g.col0
    c.group_scale_off[i] = g.scale_offset
  for d in range(net.D):
    c.fourier_degree[d] = int(net.fourier_degrees[d])
    c.input_scale[d] = float(np.float32(net.input_scales[d]))
  c.n_freqs = int(net.freqs.size)
  for j in range(net.freqs.size):
    c.freq[j] = float(net.freqs[j])
    c.harmonic[j] = float(net.harmonics[j])
  c.n_interact = int(net.interactions.shape[0])
  for k in range(c.n_interact):
    c.interact[k][0] = int(net.interactions[k, 0])
    c.interact[k][1] = int(net.interactions[k, 1])
  c.off_log_noise_scale = net.offset('log_noise_scale')
  c.off_shape = net.offset('shape')
  c.off_inflated = net.offset('inflated_loc_probs')
  for l in range(net.depth + 1):
    c.off_bias[l] = net.offset(f'Dense_{l}/bias')
    c.off_kernel[l] = net.offset(f'Dense_{l}/kernel')
  for l in range(net.depth):
    c.off_layer_scale[l] = net.offset(f'inv_sp_layer_scale{l}')
  c.off_output_scale = net.offset('inv_sp_output_scale')
  c.off_lsa = net.offset('log_scale_adjustment')
  c.off_act_weight = net.offset('logit_activation_weight')
  c.n_rows, c.batch = int(n_rows), int(batch)
  c.members, c.member_offset = int(members), int(member_offset)
  c.vi_samples = int(vi_samples)
  c.forward_only = 1 if forward_only else 0
  if pipeline not in PIPELINE and pipeline not in PIPELINE.values():
    raise ValueError(f'pipeline must be one of {sorted(PIPELINE)} (got {pipeline!r})')
  c.pipeline = PIPELINE.get(pipeline, pipeline)
  c.learning_rate = float(learning_rate)
  c.prior_weight = float(prior_weight)
  c.kl_weight = float(kl_weight)
  c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
  return c
