#!/usr/bin/env python
"""Headline benchmark: ensemble MAP train-step throughput (member-steps/s).

  python bench.py --gpus N --steps K --warmup W          (N=1 default)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[1] ("C2"): synthetic
chickenpox-like grid, 522 weekly times x 20 sites, the last 52 weeks of 4
sites held out -> N = 10,232 training rows, D = 3, periods [4, 52.1775] with
harmonics [2, 10] (F = 57), width 512, depth 2, NORMAL likelihood, MAP, full
batch, lr 0.005, bf16 MFMA contractions with f32 accumulation, 64 members PER
GPU (weak scaling: members shard over ranks, no data-path collective).

One "step" = one full-batch Adam step of every member (featurise, forward,
likelihood + prior, backward, Adam) - reference inference.py:599-607.  The whole
K-step loop is enqueued by ONE C call (bnf_train) with inputs resident in HBM.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel: algorithmic FLOPs per launch / its mean HIP-event
                duration inside the timed region, vs the dense bf16 MFMA peak.
  cpu_baseline  the oracle's algorithm on torch CPU float32 with torch.bmm over members (a port of
                the reference algorithm, NOT JAX) timed on this box's host cores on a bounded
                sample (N=1 only).
  device_preheat  ms of untimed steps on a SCRATCH engine before the W warm-up steps of the measured one
                (--preheat-ms, default 100, 0 = off): after an idle period the device ramps for ~8 steps
                whatever engine runs on it (profiles/r03_device_ramp.txt); a fit is thousands of steps.
`--gpus N` without a launcher starts the N ranks itself (torch.distributed.run, 127.0.0.1) and
refuses to run when fewer than N devices are visible; the line carries `rccl_world_size` (from an
actual all_reduce), every rank's ms/step and, for N > 1, the one posterior all-gather.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

PEAK_TFLOPS = {'bf16': 2500.0, 'fp32_split': 2500.0 / 3, 'fp32': 157.3, 'fp8': 5000.0}   # dense MFMA peaks, MI355X_MICROARCH.md ('fp8': since round 6 the
# dominant kernel's W x W contractions -- 90 % of its algorithmic FLOPs at C2 -- run on the block-scaled fp8 MFMA, so the line is
# priced against the dense fp8 peak; its layer-0 contractions and the column sums are bf16 MFMAs; 'fp32_split':
# the split-bf16 contraction issues three bf16 MFMAs per product; 'fp32': the exact f32 MFMA)

# engine kernel name (bnf_profile_read) -> device symbol (rocprofv3 Kernel_Name), {T} = element type
KERNEL_SYMBOL = {   # gemm_nt<T, epilogue, tag, wave rows, wave cols>
    'gemm_fwd_l0': 'void bnf::gemm_nt<{T}, 0, 0, 2, 2>(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_fwd': 'void bnf::gemm_nt<{T}, 0, 1, 4, 4>(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_fwd_last': 'void bnf::gemm_nt<{T}, 5, 3, 1, 8>(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_dgrad': 'void bnf::gemm_nt<{T}, 1, 2, 2, 2>(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_dgrad0': 'void bnf::gemm_nt<{T}, 2, 0, 2, 2>(bnf::GemmArgs, bnf::EpiArgs)',
    # the benchmark shape (bf16, W = 512, Fp = 64) runs the HBM-stream forms of the weight gradients
    'gemm_wgrad_l0': 'bnf::gemm_tn_skinny(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_wgrad': 'void bnf::gemm_tn_ring<1, true',   # (+ the wave count)
    'last_bwd': 'void bnf::k_last_bwd<{T}>',
    'panel_fwd_bwd': 'void bnf::k_panel_fwd_bwd<8, 4, true',   # (+ the DEEP flag: false at the benchmark depth)
}


def kernel_source_sha16():
  """sha256 (first 16 hex digits) over the engine's device sources: the committed counter files carry the value they
  were measured with (`_meta.kernel_source_sha16`, written by scripts/pmc_summary.py / counter_summary.py on the GPU
  box); a file taken with other kernel sources is STALE and is not quoted."""
  import hashlib
  h = hashlib.sha256()
  d = os.path.join(ROOT, 'bayesnf_amd', 'csrc')
  for name in sorted(os.listdir(d)):
    if name.endswith(('.h', '.hip')):
      with open(os.path.join(d, name), 'rb') as f:
        h.update(name.encode() + b'\0' + f.read() + b'\0')
  return h.hexdigest()[:16]


def _committed_counters(fname, kernel, dtype, members):
  """-> (record of `kernel` | None, provenance dict).  None + the reason when there is no measurement of THIS build:
  file absent, other member count, kernel symbol not in it, or taken with other kernel sources (stale)."""
  rel = os.path.join('profiles', fname)
  path = os.path.join(ROOT, rel)
  src = {'file': rel, 'measured': 'separate rocprofv3 --pmc passes of this command, committed; NOT re-measured in this run'}
  if members != 64 or kernel not in KERNEL_SYMBOL or dtype == 'fp8':
    return None, dict(src, used=False, why='no committed measurement for this kernel / member count / dtype')
  if not os.path.exists(path):
    return None, dict(src, used=False, why='file absent')
  with open(path) as f:
    table = json.load(f)
  meta = table.get('_meta') or {}
  src.update({k: meta[k] for k in ('commit', 'kernel_source_sha16', 'taken') if k in meta})
  here = kernel_source_sha16()
  if meta.get('kernel_source_sha16') != here:
    return None, dict(src, used=False, why=f'stale: taken with kernel sources {meta.get("kernel_source_sha16")}, this build is {here}')
  sym = KERNEL_SYMBOL[kernel].format(T='bnf::bf16_t' if dtype in ('bf16', 'fp8') else 'float').split('(')[0]
  for name, rec in table.items():
    if name != '_meta' and name.startswith(sym):
      return rec, dict(src, used=True, symbol=name)
  return None, dict(src, used=False, why=f'no kernel symbol starting with {sym!r} in the file')


def sq_counters(kernel, dtype, members):
  """Per-launch SQ counters of `kernel` from the committed rocprofv3 --pmc passes of this same command
  (profiles/sq_counters.json, made by scripts/gpu_counters.sh + scripts/counter_summary.py) + provenance."""
  return _committed_counters('sq_counters.json', kernel, dtype, members)


def issue_floors(ctr, launch_us, n_simd=1024):
  """How far the two issue-bound pipes of the dominant kernel are from saturation, from its SQ counters:
  every SIMD issues at most one VALU-class instruction per 4 cycles (MI355X_MICROARCH.md: 8 issue slots per
  32-cycle MFMA), a 32x32x16 bf16 MFMA holds the matrix pipe for 32 cycles.  Clock = GRBM_GUI_ACTIVE over the
  kernel's duration in the counter pass when both are there, else 2.0 GHz (what this kernel sustains)."""
  if not ctr or 'SQ_INSTS_VALU' not in ctr:
    return None
  ghz = 2.0
  out = {'valu_insts_per_launch': ctr['SQ_INSTS_VALU'], 'mfma_insts_per_launch': ctr.get('SQ_INSTS_MFMA'),
         'assumed_clock_ghz': ghz,
         'valu_issue_floor_us': ctr['SQ_INSTS_VALU'] / n_simd * 4.0 / (ghz * 1e3)}
  if ctr.get('SQ_INSTS_MFMA'):
    out['mfma_pipe_floor_us'] = ctr['SQ_INSTS_MFMA'] / n_simd * 32.0 / (ghz * 1e3)
    out['valu_per_mfma'] = ctr['SQ_INSTS_VALU'] / ctr['SQ_INSTS_MFMA']
  out['valu_issue_frac_of_launch'] = out['valu_issue_floor_us'] / launch_us
  return out


def pmc_traffic(kernel, dtype, members):
  """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this
  same command (profiles/pmc_traffic.json, made by scripts/gpu_pmc.sh; FETCH_SIZE and
  WRITE_SIZE in separate passes, 2*FETCH + WRITE KiB, see scripts/pmc_summary.py) + provenance.
  None when no measurement of this build is committed."""
  rec, src = _committed_counters('pmc_traffic.json', kernel, dtype, members)
  return (rec['hbm_bytes'] if rec else None), src


def synthetic_grid(seed=1234):
  """C2 data: x=(t, lat, lon), y = 3 sin(2 pi t/p1) + sin(2 pi t/p2)
  + 2 lat lon + 0.5 N(0,1).  Returns train X (N,3) float64, y (N,), input scales."""
  rng = np.random.default_rng(seed)
  T, S = 522, 20
  lat, lon = rng.uniform(-1, 1, S), rng.uniform(-1, 1, S)
  lat = (lat - lat.mean()) / lat.std()
  lon = (lon - lon.mean()) / lon.std()
  tt, ss = np.meshgrid(np.arange(T, dtype=np.float64), np.arange(S), indexing='ij')
  tt, ss = tt.ravel(), ss.ravel()
  keep = ~((ss < 4) & (tt >= T - 52))
  t, s = tt[keep], ss[keep]
  X = np.stack([t, lat[s], lon[s]], axis=1)
  p1, p2 = 4.0, 52.1775
  y = (3 * np.sin(2 * np.pi * t / p1) + np.sin(2 * np.pi * t / p2) + 2 * lat[s] * lon[s] +
       0.5 * rng.standard_normal(t.size))
  return X, y, np.array([T - 1.0, 1.0, 1.0])


MODEL_KW = dict(width=512, depth=2, fourier_degrees=[5, 5, 5], interactions=[],
                seasonality_periods=[4.0, 52.1775], num_seasonal_harmonics=[2, 10])


_CPU_BARRIER = None


def _cpu_pool_init(barrier):
  global _CPU_BARRIER
  _CPU_BARRIER = barrier


def _numa_cpu_lists():
  """[[cpu ids of node 0], [node 1], ...] from sysfs; one list with every allowed cpu when unreadable."""
  allowed = sorted(os.sched_getaffinity(0))
  nodes = []
  try:
    for d in sorted(os.listdir('/sys/devices/system/node')):
      if not d.startswith('node') or not d[4:].isdigit():
        continue
      cpus = []
      for part in open(f'/sys/devices/system/node/{d}/cpulist').read().strip().split(','):
        lo, _, hi = part.partition('-')
        cpus += list(range(int(lo), int(hi or lo) + 1))
      cpus = [c for c in cpus if c in allowed]
      if cpus:
        nodes.append(cpus)
  except (OSError, ValueError):
    nodes = []
  return nodes or [allowed]


def _pin_worker(k, nt, n_workers):
  """Worker k of the side-by-side layout (n_workers > 1) gets `nt` cpus of ONE NUMA node, strided over the
  node's physical cores so that every worker still reaches every CCD's L3 and fabric link (round 2's unpinned
  8 x 16 layout was slower than one 16-thread process: threads migrated and pages sat on the other socket;
  pinning each worker to CONSECUTIVE cores was worse still -- 4 x 32: 8.2 against 23.9 member-steps/s for one
  unpinned 32-thread process -- because two CCDs' links then carry a worker's whole stream).  The
  single-process sweep stays unpinned.  No-op where affinity cannot be set."""
  if n_workers <= 1:
    return
  try:
    nodes = _numa_cpu_lists()
    # sysfs lists a node's physical cores first and their SMT siblings second: keep the first half when the
    # node has at least twice the cpus the workers on it need
    per_node = -(-n_workers // len(nodes))
    phys = [c[:len(c) // 2] if len(c) // 2 >= per_node * nt else c for c in nodes]
    n, j = k % len(nodes), k // len(nodes)
    mine = phys[n][j::per_node][:nt]
    if len(mine) == nt:
      os.sched_setaffinity(0, set(mine))
  except (OSError, AttributeError, ZeroDivisionError):
    pass


def _cpu_worker(args):
  """One worker of the CPU baseline: `members` members x `steps` timed full-batch steps on `nt` threads."""
  X, y, input_scales, members, steps, nt, seed, n_workers = args
  import ctypes
  import torch
  from oracle import bnf_oracle as O
  from oracle.torch_baseline import TorchStep
  try:   # serve the 168 MB temporaries from the heap, not from fresh mmap'd (page-faulting) regions
    libc = ctypes.CDLL('libc.so.6')
    libc.mallopt(-3, 1 << 30)   # M_MMAP_THRESHOLD
    libc.mallopt(-1, 1 << 30)   # M_TRIM_THRESHOLD
  except OSError:
    pass
  _pin_worker(seed, nt, n_workers)
  torch.set_num_threads(nt)
  model = O.Model(input_scales=input_scales, **MODEL_KW)
  rng = np.random.default_rng(seed)
  theta0 = O.map_init(model, y, rng.standard_normal((members, model.P)).clip(-2, 2), dtype=np.float32)
  ts = TorchStep(model, X, y, lr=0.005)
  ts.train(theta0, 1)                                    # warm-up
  if _CPU_BARRIER is not None:
    _CPU_BARRIER.wait(timeout=300)                       # every process starts its timed steps together
  t0 = time.time()
  _, losses = ts.train(theta0, steps)
  t1 = time.time()
  if not np.all(np.isfinite(losses)):
    raise RuntimeError('non-finite loss in the CPU baseline')
  return t0, t1


def cpu_baseline(X, y, input_scales, members=8, steps=3):
  """The oracle's train step on the host cores: torch CPU float32, members batched with torch.bmm,
  hand-derived backward (oracle/torch_baseline.py, checked against the numpy oracle in
  tests/test_torch_baseline.py).  A bounded sample of the bench workload, same C2 inputs, tuned
  like a baseline should be: (1) one process, `members` members x `steps` timed steps, for a few
  intra-op thread counts (on the 2 x 64-core host of the GPU box 16 threads beat 128 by 6x: the
  element-wise passes do not scale); (2) with the best count T, cores / T processes side by side,
  each with its own `members` members (ensemble members are independent) -- the reported value is
  all members x steps / the span from the first start to the last finish."""
  import multiprocessing as mp
  import torch
  n_max = torch.get_num_threads()
  ctx = mp.get_context('spawn')
  tried = {}
  with ctx.Pool(1) as pool:
    for nt in sorted({n_max, max(1, n_max // 2), max(1, n_max // 4), min(n_max, 16), min(n_max, 8)}, reverse=True):
      t0, t1 = pool.map_async(_cpu_worker, [(X, y, input_scales, members, steps, nt, 0, 1)]).get(timeout=900)[0]
      tried[nt] = round(members * steps / (t1 - t0), 3)
  nt = max(tried, key=tried.get)
  procs = max(1, n_max // nt)
  value, cores = tried[nt], nt
  if procs > 1:
    try:   # (a worker that dies must not hang the bench: bounded waits, and the single-process figure stands)
      with ctx.Pool(procs, initializer=_cpu_pool_init, initargs=(ctx.Barrier(procs),)) as pool:
        spans = pool.map_async(_cpu_worker, [(X, y, input_scales, members, steps, nt, k, procs) for k in range(procs)],
                               chunksize=1).get(timeout=900)
      span = max(t for _, t in spans) - min(t for t, _ in spans)
      v = procs * members * steps / span
      tried[f'{procs}x{nt}'] = round(v, 3)
      if v > value:
        value, cores = v, procs * nt
    except Exception as exc:  # pylint: disable=broad-except
      tried[f'{procs}x{nt}'] = f'failed: {type(exc).__name__}'
  cpu = 'unknown'
  try:
    with open('/proc/cpuinfo') as f:
      cpu = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
  except Exception:  # pylint: disable=broad-except
    pass
  model_flops = 6.0 * len(y) * (57 * 512 + 512**2 + 512)
  return dict(value=value, unit='member-steps/s', cores=int(cores),
              kind='port', cpu=cpu, host_cores=os.cpu_count(), member_steps_per_s_by_threads=tried,
              achieved_tflops=model_flops * value / 1e12,
              sample=f'{members} members x {steps} full-batch steps per process (N={len(y)}, W=512, depth 2) after 1 warm-up, '
                     'torch CPU float32 + torch.bmm port of the oracle (oracle/torch_baseline.py), not JAX; '
                     'best of the thread counts / process layouts in member_steps_per_s_by_threads')


# ---------------------------------------------------------------------------------------------
# one process per GPU: launcher, process group, collectives (shared by the real run and the CPU self-test)
# ---------------------------------------------------------------------------------------------
def _free_port():
  import socket
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    return sk.getsockname()[1]


def devices_visible():
  import torch
  return torch.cuda.device_count() if torch.cuda.is_available() else 0


def require_devices(n):
  visible = devices_visible()
  if visible < n:
    raise SystemExit(f'[bench] --gpus {n} requested but devices visible: {visible}; refusing to run '
                     'a smaller job under that label')


def self_launch(args, argv):
  """`python bench.py --gpus N` without a launcher: start N ranks of this file under
  torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and pass their output through.
  Fails loudly when fewer than N devices are visible (never measures 1 GPU and calls it N)."""
  import subprocess
  if not args.selftest_cpu:
    require_devices(args.gpus)
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
  return subprocess.call(cmd, env=env)


def init_group(selftest):
  """-> (world, rank, device).  backend nccl (= RCCL over xGMI) on GPUs, gloo for the CPU self-test."""
  import torch
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world > 1 and not torch.distributed.is_initialized():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if selftest:
      torch.distributed.init_process_group(backend='gloo')
    else:
      from bayesnf_amd import distributed
      distributed.maybe_init_from_env()
  rank = torch.distributed.get_rank() if world > 1 else 0
  if selftest:
    return world, rank, torch.device('cpu')
  from bayesnf_amd import distributed
  return world, rank, torch.device(f'cuda:{distributed.local_device_index()}')


def collective_world_size(world, device):
  """Number of ranks an actual all_reduce saw (1 without a process group)."""
  import torch
  if world == 1:
    return 1
  one = torch.ones(1, dtype=torch.float32, device=device)
  torch.distributed.all_reduce(one)
  return int(round(one.item()))


def gather_rank_times(world, device, elapsed):
  """-> list of every rank's elapsed seconds (one all_gather of a scalar)."""
  import torch
  if world == 1:
    return [elapsed]
  mine = torch.tensor([elapsed], dtype=torch.float64, device=device)
  out = torch.empty(world, dtype=torch.float64, device=device)
  torch.distributed.all_gather_into_tensor(out, mine)
  return [float(v) for v in out.cpu()]


class _SelftestCommLib:
  """CPU self-test stand-in for the three communicator entry points of libbnf_hip.so (include/bnf.h:
  bnf_comm_unique_id / bnf_comm_create / bnf_allgather): records what `_native.allgather` hands over --
  the 128-byte id every rank must agree on, world / rank, buffer addresses and byte counts -- and moves the
  bytes with the gloo group, so tests/test_bench_launcher.py covers the `--gather cabi` plumbing without a GPU."""

  def __init__(self):
    self.calls = []
    self.ident = None

  def bnf_comm_unique_id(self, buf):
    buf.raw = bytes((7 * i + 3) % 251 for i in range(128))
    return 0

  def bnf_comm_create(self, raw, world, rank, device, out):
    self.ident = bytes(raw.raw)
    self.calls.append(('create', world, rank, device))
    return 0

  def bnf_allgather(self, comm, send, recv, nbytes, stream):
    import ctypes
    import torch
    n = int(nbytes.value)
    world = torch.distributed.get_world_size()
    src = torch.frombuffer((ctypes.c_char * n).from_address(send.value), dtype=torch.uint8)
    dst = torch.frombuffer((ctypes.c_char * (n * world)).from_address(recv.value), dtype=torch.uint8)
    torch.distributed.all_gather_into_tensor(dst, src)
    self.calls.append(('allgather', n))
    return 0

  def bnf_comm_available(self):
    return 0

  # one process, several devices (--launcher inproc): the communicator set and the grouped all-gather
  def bnf_comm_create_local(self, n, devices, out):
    self.calls.append(('create_local', int(n), [int(devices[i]) for i in range(int(n))]))
    for i in range(int(n)):
      out[i] = 1000 + i
    return 0

  def bnf_allgather_group(self, n, comms, send, recv, nbytes, streams):
    import ctypes
    nb = int(nbytes.value)
    for i in range(int(n)):          # every local rank receives every block, rank-major
      for j in range(int(n)):
        ctypes.memmove(recv[i] + j * nb, send[j], nb)
    self.calls.append(('allgather_group', int(n), nb))
    return 0


def gather_posterior(world, rank, device, local_means, impl, lib=None):
  """The job's only data collective (reference inference.py:452,486-492: pmap's output gather):
  every rank's device-resident predictive means (E_local, R) -> (world * E_local, R) on every rank
  with ONE all-gather over RCCL / xGMI -- impl 'cabi': `bnf_allgather` of the engine library (the C ABI a
  non-torch host would call; librccl resolved inside libbnf_hip.so), 'torch': all_gather_into_tensor of
  the nccl (= RCCL) process group.  -> (tensor, ms)."""
  import torch
  local_means = local_means.contiguous()
  if world == 1:
    return local_means, 0.0
  out = torch.empty((world * local_means.shape[0],) + tuple(local_means.shape[1:]), dtype=local_means.dtype,
                    device=device)
  if impl == 'cabi':   # communicator set-up (id broadcast, ncclCommInitRank) outside the timed collective
    from bayesnf_amd import _native
    warm_s = torch.zeros(8, dtype=torch.float32, device=device)
    warm_r = torch.zeros(8 * world, dtype=torch.float32, device=device)
    _native.allgather(warm_s, warm_r, world, rank, lib=lib)
  if device.type == 'cuda':
    torch.cuda.synchronize(device)
  torch.distributed.barrier()
  t0 = time.perf_counter()
  if impl == 'cabi':
    _native.allgather(local_means, out, world, rank, lib=lib)
  else:
    torch.distributed.all_gather_into_tensor(out, local_means)
  if device.type == 'cuda':
    torch.cuda.synchronize(device)
  return out, (time.perf_counter() - t0) * 1e3


def gather_checksums(world, device, local_means):
  """Every rank's own checksum of what it contributed (float64 sum), gathered separately from the
  payload: rank 0 compares them with the sums of the blocks the posterior gather delivered."""
  import torch
  mine = local_means.double().sum().reshape(1).to(device)
  if world == 1:
    return [float(mine.item())]
  out = torch.empty(world, dtype=torch.float64, device=device)
  torch.distributed.all_gather_into_tensor(out, mine)
  return [float(v) for v in out.cpu()]


def gather_posterior_inproc(parts, lib=None):
  """--launcher inproc: one tensor per local device -> (n * E_local, R): ONE grouped RCCL all-gather over the local
  communicator set (`_native.allgather_local`; `lib`: the CPU self-test's stand-in), peer copies when RCCL cannot
  serve the device list.  -> (tensor on the first device, ms, impl, error | None)."""
  import torch
  from bayesnf_amd import _native, distributed
  parts = [p.contiguous() for p in parts]
  n = len(parts)
  err = None
  on_gpu = parts[0].is_cuda
  if lib is not None or on_gpu:
    try:
      recvs = [torch.empty((n,) + tuple(p.shape), dtype=p.dtype, device=p.device) for p in parts]
      if on_gpu:
        warm_s = [torch.zeros(8, dtype=torch.float32, device=p.device) for p in parts]   # communicator set-up outside the timing
        warm_r = [torch.zeros(8 * n, dtype=torch.float32, device=p.device) for p in parts]
        _native.allgather_local(warm_s, warm_r, lib=lib)
        for p in parts:
          torch.cuda.synchronize(p.device)
      t0 = time.perf_counter()
      _native.allgather_local(parts, recvs, lib=lib)
      if on_gpu:
        for p in parts:
          torch.cuda.synchronize(p.device)
      ms = (time.perf_counter() - t0) * 1e3
      return recvs[0].view((n * parts[0].shape[0],) + tuple(parts[0].shape[1:])), ms, 'rccl-group', None
    except (RuntimeError, OSError) as exc:
      err = f'{type(exc).__name__}: {exc}'[:300]
  t0 = time.perf_counter()
  out = torch.cat([p.to(parts[0].device) for p in parts], dim=0)
  if on_gpu:
    torch.cuda.synchronize(parts[0].device)
  return out, (time.perf_counter() - t0) * 1e3, 'peer-copies', err


def run_rank(args, rank, device, E, data, sync):
  """One rank's share of the job (one process per GPU, or one host thread per GPU under --launcher inproc):
  engine set-up, device preheat, W warm-up steps, then EXACTLY K timed steps between two `sync()` rendezvous.
  -> dict(elapsed, prof, dominant, final_loss, net, eng, local_means | None)."""
  import torch
  X, y, input_scales = data
  if args.selftest_cpu:
    # stand-in for the engine: a fixed amount of host work per "step", rank-dependent means
    work = torch.randn(256, 256)
    def run_steps(k):
      for _ in range(k):
        torch.mm(work, work)
    run_steps(args.warmup)
    sync()
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync()
    return dict(elapsed=time.perf_counter() - t0, prof=None, dominant='selftest', final_loss=0.0, net=None, eng=None,
                local_means=torch.full((E, 16), float(rank), device=device), device=str(device))
  from bayesnf_amd.engine import Engine
  from bayesnf_amd.spec import NetSpec
  net = NetSpec(input_scales=input_scales, **MODEL_KW)
  eng = Engine(net, mode='map', X=X, y=y, members=E, member_offset=rank * E, seed=0,
               learning_rate=0.005, prior_weight=1.0, compute_dtype=args.dtype, device_index=device.index)
  eng.init_params(float(np.log(np.nanstd(y) / 2)))
  # the device ramps for ~8 steps (16 ms) after an idle period, whatever engine runs on it: a fresh engine's steps
  # 2..8 take 2.6 -> 2.03 ms, a second engine's in the same process 2.11 -> 2.0 (profiles/r03_device_ramp.txt)
  if args.preheat_ms > 0:
    scratch = Engine(net, mode='map', X=X, y=y, members=E, member_offset=rank * E, seed=1,
                     learning_rate=0.005, prior_weight=1.0, compute_dtype=args.dtype, device_index=device.index)
    scratch.init_params(0.0)
    t_h = time.perf_counter()
    ep_h = 0
    while (time.perf_counter() - t_h) * 1e3 < args.preheat_ms:
      scratch.train(ep_h, 8)
      torch.cuda.synchronize(device)
      ep_h += 8
    scratch.close()
  # ---- warm-up (also finds the dominant kernel) --------------------------------
  eng.profile('*')
  eng.train(0, max(args.warmup, 1))
  sync()
  warm = eng.profile_read()
  eng.profile(None)
  total_ms = {k: v['avg_ms'] * v['calls'] for k, v in warm.items()}
  dominant = max(total_ms, key=total_ms.get)
  if args.profile_all and rank == 0:
    for k, v in sorted(warm.items(), key=lambda kv: -total_ms[kv[0]]):
      tf = v['flops'] / (v['avg_ms'] * 1e-3) / 1e12 if v['flops'] else 0.0
      print(f'[bench] {k:16s} avg {v["avg_ms"]*1e3:9.1f} us  x{v["calls"]:4d}  '
            f'{tf:8.1f} TFLOP/s', file=sys.stderr)
  # ---- timed region: exactly K steps, events only around the dominant kernel ---
  eng.profile(dominant)
  sync()
  t0 = time.perf_counter()
  losses = eng.train(args.warmup, args.steps)
  sync()
  elapsed = time.perf_counter() - t0
  prof = eng.profile_read()
  eng.profile(None)
  final_loss = float(losses[:, -1].mean().item())
  if not np.isfinite(final_loss):
    raise RuntimeError('non-finite training loss in the benchmark run')
  return dict(elapsed=elapsed, prof=prof, dominant=dominant, final_loss=final_loss, net=net, eng=eng, local_means=None,
              device=str(device))


def side_dtype_lines(args, device, E, X, y, input_scales, flops_step):
  """After the timed region, N = 1 only: the same C2 step under the engine's other arithmetics, short runs (8 warm-up + 20
  timed steps each, a fresh engine, the device already warm) -- reported beside the headline, never as it: 'fp8' (BASELINE
  configs[4]'s arithmetic: fp8 MFMA contractions + fp8 operand storage) and 'fp32_split' (what the estimators run when no
  compute_dtype is given)."""
  import torch
  from bayesnf_amd.engine import Engine
  from bayesnf_amd.spec import NetSpec
  net = NetSpec(input_scales=input_scales, **MODEL_KW)
  out = {}
  for dt in ('fp8', 'fp32_split'):
    if dt == args.dtype:
      continue
    try:
      eng = Engine(net, mode='map', X=X, y=y, members=E, seed=0, learning_rate=0.005, prior_weight=1.0, compute_dtype=dt,
                   device_index=device.index)
      eng.init_params(float(np.log(np.nanstd(y) / 2)))
      eng.train(0, 8)
      torch.cuda.synchronize(device)
      t0 = time.perf_counter()
      eng.train(8, 20)
      torch.cuda.synchronize(device)
      dt_s = time.perf_counter() - t0
      eng.close()
      out[dt] = {'ms_per_step': dt_s * 1e3 / 20, 'member_steps_per_s': E * 20 / dt_s, 'steps': 20,
                 'algorithmic_tflops': flops_step * 20 / dt_s / 1e12}
    except Exception as exc:   # pylint: disable=broad-except
      out[dt] = {'error': f'{type(exc).__name__}: {exc}'[:200]}
  return out


def predictive_means(args, r, device, E, X):
  """this rank's members on the first 1024 training rows (forward-only handle): what the posterior gather carries"""
  import torch
  from bayesnf_amd.engine import Engine
  fwd = Engine(r['net'], members=E, forward_only=True, row_capacity=1024, compute_dtype=args.dtype, device_index=device.index)
  Xd = torch.from_numpy(np.ascontiguousarray(X[:1024], dtype=np.float32)).to(device)
  means, _ = fwd.forward(r['eng'].params.view(E, r['net'].P), Xd)
  torch.cuda.synchronize(device)
  fwd.close()
  return means


def main(argv=None):
  argv = list(sys.argv[1:] if argv is None else argv)
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=30)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--members-per-gpu', type=int, default=64)
  ap.add_argument('--strong', action='store_true',
                  help='strong scaling: --members-per-gpu is the size of the WHOLE ensemble, split evenly over the '
                       'ranks (default: weak scaling, that many members on every GPU)')
  ap.add_argument('--launcher', default='torchrun', choices=['torchrun', 'inproc'],
                  help='N > 1: one process per GPU under torch.distributed.run (default; what the driver starts), or '
                       "ONE process driving every GPU from one host thread each (`inproc`: the reference's own shape, "
                       'jax.pmap over jax.local_devices(); the gather is one grouped RCCL all-gather)')
  ap.add_argument('--check', action='store_true',
                  help='pre-flight for N > 1: create the communicator(s), all-gather 1 KiB per rank, verify, print one '
                       'JSON line and exit -- an RCCL set-up failure costs seconds, not the lease')
  ap.add_argument('--gather', default=None, choices=['cabi', 'torch'],
                  help="posterior gather through bnf_allgather of the engine library (default on GPUs; env "
                       "BNF_GATHER) or torch.distributed's all_gather_into_tensor")
  ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'fp32_split', 'fp8'],
                  help="bf16 (the headline), fp32 (the parity arithmetic), fp8 = bf16 contractions with fp8 operand storage for "
                       'the weight-gradient streams (BASELINE configs[4])')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--preheat-ms', type=float, default=100.0,
                  help='bring the DEVICE to the power / clock state a long fit runs in: this many ms of untimed steps on a '
                       'scratch engine (other parameters, other buffers) before the W warm-up steps of the measured engine; '
                       '0 = off.  A 20-step sample that starts on an idle device reads ~3 %% above the 200-step rate '
                       '(profiles/r03z_bench_200_steps.txt, r03_device_ramp.txt)')
  ap.add_argument('--profile-all', action='store_true',
                  help='also print the per-kernel HIP-event table to stderr')
  ap.add_argument('--selftest-cpu', action='store_true',
                  help='no GPU: exercise launcher, process group (gloo), timing reduction, posterior gather and '
                       'the JSON line with a stand-in step (tests/test_bench_launcher.py)')
  args = ap.parse_args(argv)
  inproc = args.launcher == 'inproc' and args.gpus > 1 and 'WORLD_SIZE' not in os.environ

  if args.gpus > 1 and not inproc and 'WORLD_SIZE' not in os.environ:
    sys.exit(self_launch(args, argv))

  import torch
  import threading
  visible = 0 if args.selftest_cpu else devices_visible()
  if inproc:
    # ONE process, one engine + one host thread per device (bayesnf_amd.distributed.run_shards); no process group
    from bayesnf_amd import distributed
    if not args.selftest_cpu:
      require_devices(args.gpus)
    world, rank = args.gpus, 0
    devs = (list(range(world)) if args.selftest_cpu else
            (distributed.local_devices() if os.environ.get('BNF_DEVICES') else list(range(world)))[:world])
    if len(devs) != world:
      raise SystemExit(f'[bench] --launcher inproc --gpus {world}: BNF_DEVICES names {len(devs)} device(s)')
    devices = [torch.device('cpu') if args.selftest_cpu else torch.device(f'cuda:{d}') for d in devs]
    device = devices[0]
    seen = world
  else:
    world, rank, device = init_group(args.selftest_cpu)
    if args.gpus != world:
      raise SystemExit(f'[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    seen = collective_world_size(world, device)
    if seen != world:
      raise SystemExit(f'[bench] all_reduce saw {seen} ranks, expected {world}')
    devices = [device]

  gather_impl = args.gather or os.environ.get('BNF_GATHER') or ('torch' if args.selftest_cpu else 'cabi')
  comm_lib = _SelftestCommLib() if (args.selftest_cpu and (gather_impl == 'cabi' or inproc)) else None

  if args.check:
    # ---- pre-flight: communicator(s) + a 1 KiB all-gather, nothing else -------------------------------------
    t0 = time.perf_counter()
    if inproc:
      parts = [torch.full((1, 256), float(i), dtype=torch.float32, device=d) for i, d in enumerate(devices)]
      allm, _, impl, err = gather_posterior_inproc(parts, lib=comm_lib)
    else:
      part = torch.full((1, 256), float(rank), dtype=torch.float32, device=device)
      err = None
      try:
        allm, _ = gather_posterior(world, rank, device, part, gather_impl, lib=comm_lib)
        impl = gather_impl
      except RuntimeError as exc:   # (raised on every rank together: _native.allgather agrees first)
        if gather_impl != 'cabi':
          raise
        err, impl = f'{type(exc).__name__}: {exc}'[:300], 'torch'
        allm, _ = gather_posterior(world, rank, device, part, 'torch')
    ok = allm.view(world, -1)[:, 0].tolist() == [float(r) for r in range(world)]
    if rank == 0:
      print(json.dumps({'check': 'ok' if ok else 'FAILED', 'n_gpus': world, 'launcher': 'inproc' if inproc else 'torchrun',
                        'devices_visible': visible, 'rank_devices': [str(d) for d in devices] if inproc else None,
                        'gather_impl': impl, 'error': err, 'bytes_per_rank': 1024,
                        's': round(time.perf_counter() - t0, 3)}), flush=True)
    if world > 1 and not inproc:
      torch.distributed.barrier()
      torch.distributed.destroy_process_group()
    sys.exit(0 if ok else 1)

  data = synthetic_grid()
  X, y, input_scales = data
  E = args.members_per_gpu
  if args.strong:
    if E % world:
      raise SystemExit(f'[bench] --strong: {E} members do not split evenly over {world} ranks')
    E //= world

  gather = None
  if inproc:
    # every device's thread meets the others on both sides of the timed region; its own device is synchronised first
    meet = threading.Barrier(world)
    def make_sync(dev):
      def sync():
        if dev.type == 'cuda':
          torch.cuda.synchronize(dev)
        meet.wait(timeout=1800)
      return sync
    shards = [distributed.Shard(i, devs[i]) for i in range(world)]
    results = distributed.run_shards(lambda sh: run_rank(args, sh.index, devices[sh.index], E, data, make_sync(devices[sh.index])),
                                     shards)
    r0 = results[0]
    rank_s = [r['elapsed'] for r in results]
    rank_devices = [r['device'] for r in results]
    parts = [(r['local_means'] if r['local_means'] is not None else predictive_means(args, r, devices[i], E, X))
             for i, r in enumerate(results)]
    sums = [float(p.double().sum().item()) for p in parts]
    allm, ms, impl, err = gather_posterior_inproc(parts, lib=comm_lib)
    gather_impl = impl
    cabi_error = err
  else:
    def sync():
      if device.type == 'cuda':
        torch.cuda.synchronize(device)
      if world > 1:
        torch.distributed.barrier()
        if device.type == 'cuda':
          torch.cuda.synchronize(device)
    r0 = run_rank(args, rank, device, E, data, sync)
    results = [r0]
    rank_s = gather_rank_times(world, device, r0['elapsed'])
    rank_devices = None
    if world > 1:
      ords = torch.tensor([device.index if device.type == 'cuda' else -1], dtype=torch.int64, device=device)
      allo = torch.empty(world, dtype=torch.int64, device=device)
      torch.distributed.all_gather_into_tensor(allo, ords)
      rank_devices = [('cpu' if v < 0 else f'cuda:{int(v)}') for v in allo.cpu()]
      local_means = r0['local_means'] if r0['local_means'] is not None else predictive_means(args, r0, device, E, X)
      sums = gather_checksums(world, device, local_means)
      cabi_error = None
      try:
        allm, ms = gather_posterior(world, rank, device, local_means, gather_impl, lib=comm_lib)
      except RuntimeError as exc:
        if gather_impl != 'cabi':
          raise
        # the C-ABI communicator could not be set up on this node (every rank raised together: `_native.allgather`
        # agrees across ranks before it raises): say so in the line and gather with the process group instead (the
        # timed region is already over; the collective is not part of `value`)
        cabi_error, gather_impl = f'{type(exc).__name__}: {exc}'[:300], 'torch'
        allm, ms = gather_posterior(world, rank, device, local_means, gather_impl)
      parts = [local_means]
  elapsed = max(rank_s)

  if world > 1:
    got = [float(v) for v in allm.view(world, -1).double().sum(dim=1).cpu()]
    names = {'cabi': 'bnf_allgather (C ABI of libbnf_hip.so -> ncclAllGather, RCCL)',
             'rccl-group': 'bnf_comm_create_local + bnf_allgather_group (C ABI of libbnf_hip.so -> ncclCommInitAll, grouped ncclAllGather, RCCL)',
             'peer-copies': 'peer copies to the first device (RCCL could not serve the device list)'}
    gather = {'impl': names.get(gather_impl) or 'torch.distributed.all_gather_into_tensor (backend %s)' % torch.distributed.get_backend(),
              'shape': list(allm.shape), 'bytes_per_rank': parts[0].numel() * parts[0].element_size(),
              'ms': ms, 'finite': bool(torch.isfinite(allm).all().item()),
              'rank_checksums': sums,
              'rank_checksums_ok': all(abs(a - b) <= 1e-9 * max(1.0, abs(a)) for a, b in zip(sums, got))}
    if cabi_error:
      gather['cabi_error'] = cabi_error
    if args.selftest_cpu:   # every rank's block must hold that rank's id
      blocks = allm.view(world, E, -1)[:, 0, 0].tolist()
      gather['rank_blocks_ok'] = blocks == [float(r) for r in range(world)]
      if comm_lib is not None:
        gather['cabi_calls'] = comm_lib.calls
        if comm_lib.ident is not None:
          gather['cabi_id_head'] = list(comm_lib.ident[:4])

  if rank == 0:
    total_members = E * world
    value = total_members * args.steps / elapsed
    line = {
        'metric': 'train-steps/sec x ensemble_size (member-steps/s, whole job)',
        'value': value, 'unit': 'member-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'C2 chickenpox-like MAP: N=10232 rows, D=3, F=57, width=512, '
                               'depth=2, NORMAL, full batch, lr=0.005',
                   'members_per_gpu': E, 'ensemble_size': total_members, 'parallelism':
                   f'ensemble-shard x{world} (no data-path collective)'},
        'launcher': ('inproc: one process, one host thread per GPU' if inproc else
                     ('torch.distributed.run: one process per GPU' if world > 1 else 'single process')),
        'devices_visible': visible, 'rank_devices': rank_devices,
        'rccl_world_size': seen, 'per_rank_ms_per_step': [t / args.steps * 1e3 for t in rank_s],
        'final_loss_mean': r0['final_loss'],
    }
    if gather is not None:
      line['posterior_gather'] = gather
    if args.selftest_cpu:
      line['selftest'] = True
    else:
      prof, dominant, net = r0['prof'], r0['dominant'], r0['net']
      line['device_preheat'] = {'ms': args.preheat_ms, 'what': 'untimed steps of a scratch engine (other parameters and buffers) '
                                'before the W warm-up steps of the measured engine: device power / clock state; --preheat-ms 0 = off'}
      d = prof[dominant]
      achieved = d['flops'] / (d['avg_ms'] * 1e-3) / 1e12 if d['flops'] else 0.0
      peak = PEAK_TFLOPS[args.dtype]
      flops_step = net.flops_per_member_step(len(y)) * total_members
      line['algorithmic_tflops'] = flops_step * args.steps / elapsed / 1e12
      traffic, traffic_src = pmc_traffic(dominant, args.dtype, E)
      ctr, ctr_src = sq_counters(dominant, args.dtype, E)
      line['roofline'] = {'bound': 'mfma', 'kernel': dominant, 'achieved': achieved, 'peak': peak,
                          'unit': 'TFLOP/s', 'frac': achieved / peak,
                          'traffic': traffic, 'traffic_source': traffic_src,
                          'avg_launch_us': d['avg_ms'] * 1e3, 'launches': d['calls'],
                          'flops_per_launch': d['flops'],
                          'issue_floors': issue_floors(ctr, d['avg_ms'] * 1e3), 'issue_floors_source': ctr_src}
      if world == 1 and not args.no_cpu_baseline:
        line['other_dtypes'] = side_dtype_lines(args, device, E, X, y, input_scales, flops_step)
        line['cpu_baseline'] = cpu_baseline(X, y, input_scales)
    print(json.dumps(line), flush=True)
  for r in results:
    if r['eng'] is not None:
      r['eng'].close()
  if world > 1 and not inproc:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
