"""Ensemble members are the unit of parallelism; a device never exchanges data while it trains.

The reference shards members with `jax.pmap` over `jax.local_devices()` of ONE process and never
communicates during training (/root/reference/src/bayesnf/inference.py:573-579, members per device
= ensemble_size // device_count, :365,:445).  Two ways to drive several GPUs here:

  * one process, every visible GPU (the reference's own shape; no launcher): `fit()` / `predict()`
    open one engine handle per device -- `BNF_DEVICES=0,1,...` selects / orders them, default all
    visible -- device g owns the members `[g * E/G, (g + 1) * E/G)`, every device's whole
    optimisation is enqueued from its own host thread before anything is waited for, and the results
    are assembled with the leading dims `(G, E/G)` by one grouped RCCL all-gather over a local
    communicator set (`bnf_comm_create_local` + `bnf_allgather_group`) -- the default once the device set
    has passed a time-limited 1 KiB first-use check (`group_gather_verdict`) -- or by peer copies to the
    first device (repeated ordinals, a failed check, `BNF_GATHER=peer`);
  * one process per GPU under `torch.distributed.run` (backend "nccl" == RCCL over xGMI): rank r
    owns device LOCAL_RANK and the members of global device index r; the only collective is the
    final all-gather of fitted parameters / predictive means.
"""

from __future__ import annotations

import os
import threading
from typing import Callable, NamedTuple

import numpy as np
import torch


def is_distributed() -> bool:
  return torch.distributed.is_available() and torch.distributed.is_initialized()


def local_devices() -> list[int]:
  """HIP device ordinals this process drives.  Under torch.distributed: the one of LOCAL_RANK.
  Otherwise `BNF_DEVICES` (comma separated ordinals; repeats allowed -- "0,0" runs two handles on
  one GPU) or every visible device, like `jax.local_devices()`."""
  if is_distributed():
    return [local_device_index()]
  env = os.environ.get('BNF_DEVICES', '').strip()
  if env:
    try:
      devs = [int(x) for x in env.split(',') if x.strip()]
    except ValueError:
      raise ValueError(f'BNF_DEVICES={env!r}: comma separated device ordinals expected') from None
    if not devs:
      raise ValueError(f'BNF_DEVICES={env!r} names no device')
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    bad = [d for d in devs if d < 0 or (n > 0 and d >= n)]
    if bad:
      raise ValueError(f'BNF_DEVICES={env!r}: ordinal(s) {bad} out of range, {n} device(s) visible')
    return devs
  if 'LOCAL_RANK' in os.environ:      # under a launcher, before (or without) the process group: this rank's device
    return [local_device_index()]
  return list(range(max(1, torch.cuda.device_count())))


def device_count() -> int:
  """Number of devices in the job -- the `jax.device_count()` analogue and the leading dim of every
  fitted array: the world size under torch.distributed, else the devices of this process."""
  return torch.distributed.get_world_size() if is_distributed() else len(local_devices())


def rank() -> int:
  return torch.distributed.get_rank() if is_distributed() else 0


class Shard(NamedTuple):
  index: int     # global device index: owns members [index * E/G, (index + 1) * E/G)
  device: int    # HIP ordinal in this process


def local_shards() -> list[Shard]:
  """The (global device index, local ordinal) pairs this process is responsible for."""
  if is_distributed():
    # (a gloo job without GPUs -- the CPU tests of the sharding arithmetic -- has no ordinal to offer: the engine refuses)
    return [Shard(rank(), local_device_index() if torch.cuda.device_count() > 0 else 0)]
  return [Shard(i, d) for i, d in enumerate(local_devices())]


def run_shards(fn: Callable[[Shard], object], shards: list[Shard] | None = None) -> list:
  """fn(shard) for every local shard -> results in shard order.  Several shards run from one host
  thread each: `bnf_train` enqueues a whole optimisation (minutes of kernels) and the launch queue
  throttles the enqueuing thread, so devices only overlap when each has its own thread (ctypes
  releases the GIL for the duration of the call)."""
  shards = local_shards() if shards is None else shards
  if len(shards) == 1:
    return [fn(shards[0])]
  out: list = [None] * len(shards)
  err: list = [None] * len(shards)
  if torch.cuda.is_available():
    from . import _native
    _native.load()          # once, before the threads (the loader's module global is not guarded)

  def work(i, sh):
    try:
      if torch.cuda.is_available():
        torch.cuda.set_device(sh.device)
      out[i] = fn(sh)
    except BaseException as e:   # pylint: disable=broad-except
      err[i] = e

  threads = [threading.Thread(target=work, args=(i, sh), name=f'bnf-dev{sh.device}') for i, sh in enumerate(shards)]
  for t in threads:
    t.start()
  for t in threads:
    t.join()
  for e in err:
    if e is not None:
      raise e
  return out


_gather_note = {}
_group_verdict: dict = {}        # device ordinals of a local set -> did the grouped all-gather pass its first-use check?


def _group_allgather(sends: list[torch.Tensor], recvs: list[torch.Tensor]) -> None:
  """The one grouped RCCL all-gather of the in-process mode (seam: the CPU tests put a stand-in here)."""
  from . import _native
  _native.allgather_local(sends, recvs)


def _sync(t: torch.Tensor) -> None:
  if t.is_cuda:
    torch.cuda.current_stream(t.device).synchronize()


def _distinct_gpus(parts: list[torch.Tensor]) -> bool:
  devs = [p.device for p in parts]
  return len(parts) > 1 and all(d.type == 'cuda' for d in devs) and len(set(devs)) == len(devs)


def group_gather_verdict(parts: list[torch.Tensor]) -> bool:
  """First use of a set of local devices: ONE 1 KiB grouped all-gather (the `bench.py --check` exchange) on a watchdog
  thread with a time limit (`BNF_GATHER_TIMEOUT_S`, default 20 s) -- communicator set-up, the collective, a value check
  of every block on every device.  The verdict is cached per device set: True makes the RCCL all-gather the default
  posterior gather of that set, False (exception, wrong values, or no answer in time -- the thread is then left behind
  as a daemon, it cannot be cancelled) the peer copies.  So the north star's path is the default the first time two
  GPUs exist, and a hang inside ncclCommInitAll / ncclGroupEnd cannot stall a `fit()` whose training has finished."""
  key = tuple(str(p.device) for p in parts)
  if key in _group_verdict:
    return _group_verdict[key]
  n, res = len(parts), {}

  def probe():
    try:
      sends = [torch.full((256,), float(i + 1), dtype=torch.float32, device=p.device) for i, p in enumerate(parts)]
      recvs = [torch.zeros((n, 256), dtype=torch.float32, device=p.device) for p in parts]
      for t in sends:
        _sync(t)
      _group_allgather(sends, recvs)
      for t in recvs:
        _sync(t)
      want = torch.arange(1, n + 1, dtype=torch.float32)[:, None].expand(n, 256)
      res['ok'] = all(bool(torch.equal(r.cpu(), want)) for r in recvs)
      if not res['ok']:
        res['error'] = 'grouped all-gather delivered wrong blocks'
    except BaseException as exc:   # pylint: disable=broad-except
      res['ok'], res['error'] = False, f'{type(exc).__name__}: {exc}'[:300]

  limit = float(os.environ.get('BNF_GATHER_TIMEOUT_S', '20'))
  th = threading.Thread(target=probe, name='bnf-rccl-check', daemon=True)
  th.start()
  th.join(limit)
  if th.is_alive():
    res['ok'], res['error'] = False, f'grouped all-gather check gave no answer within {limit:g} s'
  _group_verdict[key] = bool(res.get('ok'))
  _gather_note['check'] = {'devices': key, 'ok': _group_verdict[key], **({'error': res['error']} if res.get('error') else {})}
  return _group_verdict[key]


def gather_shards(parts: list[torch.Tensor]) -> torch.Tensor:
  """One tensor (...) per local shard -> (device_count, ...) : the reference's implicit pmap output
  gather (inference.py:452,486-492).  torch.distributed: one all-gather (`all_gather_stack`).  One process with
  several DISTINCT GPUs: ONE grouped RCCL all-gather over the local communicator set (`_native.allgather_local`:
  bnf_comm_create_local + bnf_allgather_group -- every device allocates a (device_count, ...) receive buffer and receives
  every block, the first device's copy is returned) once that set has passed `group_gather_verdict`'s first-use check;
  peer copies to the first shard's device otherwise (repeated ordinals, a failed or timed-out check, an RCCL exception).
  `BNF_GATHER=peer` forces the copies, `BNF_GATHER=rccl` the collective without the check; `last_gather()` says which
  ran and what the check found.  (No box with two GPUs was available to any round: the check is what stands between an
  untried communicator set-up and a user's finished fit.)"""
  if is_distributed():
    assert len(parts) == 1
    return all_gather_stack(parts[0])
  dev0 = parts[0].device
  mode = os.environ.get('BNF_GATHER', 'auto')
  if _distinct_gpus(parts) and mode != 'peer' and (mode == 'rccl' or group_gather_verdict(parts)):
    try:
      sends = [p.contiguous() for p in parts]
      for p in sends:
        _sync(p)                                            # producers ran on other host threads' streams
      recvs = [torch.empty((len(parts),) + tuple(p.shape), dtype=p.dtype, device=p.device) for p in sends]
      _group_allgather(sends, recvs)
      for p in sends:
        _sync(p)
      _gather_note['impl'] = 'rccl-group'
      return recvs[0]
    except (RuntimeError, OSError) as exc:
      _gather_note['error'] = f'{type(exc).__name__}: {exc}'[:300]
  _gather_note['impl'] = 'peer-copies'
  return torch.stack([p if p.device == dev0 else p.to(dev0) for p in parts])


def last_gather() -> dict:
  """{'impl': 'rccl-group' | 'peer-copies', 'error': ..., 'check': {...}} of the last in-process `gather_shards`."""
  return dict(_gather_note)


def local_device_index() -> int:
  n = torch.cuda.device_count()
  if n <= 0:
    raise RuntimeError(
        'bayesnf_amd needs an AMD gfx950 GPU (torch.cuda.is_available() is '
        'False) and has no CPU fallback.')
  return int(os.environ.get('LOCAL_RANK', '0')) % n


def member_range(total_members: int, world: int | None = None,
                 r: int | None = None) -> tuple[int, int]:
  """(first global member id, count) owned by rank r; floors like the
  reference (`ensemble_size // device_count`, inference.py:445)."""
  world = device_count() if world is None else world
  r = rank() if r is None else r
  per = total_members // world
  return r * per, per


def maybe_init_from_env():
  """Initialise the default process group when launched by torchrun."""
  if is_distributed() or 'RANK' not in os.environ or 'WORLD_SIZE' not in os.environ:
    return
  if int(os.environ['WORLD_SIZE']) <= 1:
    return
  backend = 'nccl' if torch.cuda.is_available() else 'gloo'
  if backend == 'nccl':
    torch.cuda.set_device(local_device_index())
  torch.distributed.init_process_group(backend=backend)


def all_gather_stack(t: torch.Tensor) -> torch.Tensor:
  """(…) per rank -> (world, …) on every rank: the posterior gather.  ONE collective into one
  preallocated device tensor (`all_gather_into_tensor`: RCCL all-gather over xGMI under backend
  nccl), no per-rank list and no stack copy.  BNF_GATHER=cabi routes it through the engine
  library's own RCCL entry point (`bnf_allgather`, include/bnf.h) instead."""
  if not is_distributed():
    return t.unsqueeze(0)
  t = t.contiguous()
  world = device_count()
  flat = t.reshape(1, -1)                  # (1, n): the collective concatenates along dim 0
  out = torch.empty((world, flat.shape[1]), dtype=t.dtype, device=t.device)
  if os.environ.get('BNF_GATHER') == 'cabi' and t.is_cuda:
    from . import _native
    _native.allgather(flat, out, world, rank())
  else:
    torch.distributed.all_gather_into_tensor(out, flat)
  return out.view((world,) + tuple(t.shape))


def all_gather_numpy(a: np.ndarray, device=None) -> np.ndarray:
  """numpy convenience wrapper around `all_gather_stack`."""
  if not is_distributed():
    return a[None]
  t = torch.from_numpy(np.ascontiguousarray(a))
  if torch.distributed.get_backend() == 'nccl':
    t = t.to(device if device is not None else f'cuda:{local_device_index()}')
  return all_gather_stack(t).cpu().numpy()
