"""One process per GPU.  Ensemble members are the unit of parallelism.

The reference shards members with `jax.pmap` over local devices and never
communicates during training (/root/reference/src/bayesnf/inference.py:573-579,
members per device = ensemble_size // device_count, :365,:445).  Here every
rank (launched by `torch.distributed.run`, backend "nccl" == RCCL over xGMI)
owns the members `[rank * E/G, (rank + 1) * E/G)`; the only collective is the
final gather of fitted parameters / predictive means.
"""

from __future__ import annotations

import os

import numpy as np
import torch


def is_distributed() -> bool:
  return torch.distributed.is_available() and torch.distributed.is_initialized()


def device_count() -> int:
  """Number of GPUs (= ranks) in the job; the `jax.device_count()` analogue."""
  return torch.distributed.get_world_size() if is_distributed() else 1


def rank() -> int:
  return torch.distributed.get_rank() if is_distributed() else 0


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
