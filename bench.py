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
  cpu_baseline  the numpy oracle (a port of the reference algorithm, NOT JAX)
                timed on this box's host cores on a bounded sample (N=1 only).
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

PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md

# engine kernel name (bnf_profile_read) -> device symbol (rocprofv3 Kernel_Name), {T} = element type
KERNEL_SYMBOL = {   # gemm_nt<T, epilogue, tag, wave rows, wave cols>
    'gemm_fwd_l0': 'void bnf::gemm_nt<{T}, 0, 0, 2, 2>(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_fwd': 'void bnf::gemm_nt<{T}, 0, 1, 4, 4>(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_fwd_last': 'void bnf::gemm_nt<{T}, 5, 3, 1, 8>(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_dgrad': 'void bnf::gemm_nt<{T}, 1, 2, 2, 2>(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_dgrad0': 'void bnf::gemm_nt<{T}, 2, 0, 2, 2>(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_wgrad_l0': 'void bnf::gemm_tn<{T}, 0, 2>(bnf::GemmArgs, bnf::EpiArgs)',
    'gemm_wgrad': 'void bnf::gemm_tn<{T}, 1, 4>(bnf::GemmArgs, bnf::EpiArgs)',
    'last_bwd': 'void bnf::k_last_bwd<{T}>',
    'fused_fwd_bwd': 'void bnf::k_fused_fwd_bwd<{T}',
    'panel_fwd_bwd': 'void bnf::k_panel_fwd_bwd<8, 4, true>(bnf::PanelArgs)',
}


def pmc_traffic(kernel, dtype, members):
  """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this
  same command (profiles/pmc_traffic.json, made by scripts/gpu_pmc.sh; FETCH_SIZE and
  WRITE_SIZE in separate passes, 2*FETCH + WRITE KiB, see scripts/pmc_summary.py).
  None when no matching measurement is committed."""
  path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
  if members != 64 or kernel not in KERNEL_SYMBOL or not os.path.exists(path):
    return None
  sym = KERNEL_SYMBOL[kernel].format(T='bnf::bf16_t' if dtype == 'bf16' else 'float')
  with open(path) as f:
    table = json.load(f)
  for name, rec in table.items():
    if name.startswith(sym):
      return rec['hbm_bytes']
  return None


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


def cpu_baseline(X, y, input_scales, members=4, steps=6):
  """Oracle train steps on the host: float32 numpy (BLAS threads = all cores); a bounded sample
  of the bench workload (about 10 s of CPU work at the default 4 members x 6 steps)."""
  from oracle import bnf_oracle as O
  model = O.Model(input_scales=input_scales, **MODEL_KW)
  rng = np.random.default_rng(0)
  theta0 = O.map_init(model, y, rng.standard_normal((members, model.P)).clip(-2, 2),
                      dtype=np.float32)
  Xf = X.astype(np.float32)
  yf = y.astype(np.float32)
  O.train_map(model, theta0, Xf, yf, lr=0.005, num_epochs=1, dtype=np.float32)   # warm-up
  t0 = time.perf_counter()
  O.train_map(model, theta0, Xf, yf, lr=0.005, num_epochs=steps, dtype=np.float32)
  dt = time.perf_counter() - t0
  try:
    from threadpoolctl import threadpool_info
    threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
  except Exception:  # pylint: disable=broad-except
    threads = os.cpu_count() or 1
  return dict(value=members * steps / dt, unit='member-steps/s', cores=int(threads),
              kind='port',
              sample=f'{members} members x {steps} full-batch steps (N={len(y)}, W=512, depth 2), '
                     'numpy float32 oracle (oracle/bnf_oracle.py), not JAX')


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=30)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--members-per-gpu', type=int, default=64)
  ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--profile-all', action='store_true',
                  help='also print the per-kernel HIP-event table to stderr')
  args = ap.parse_args()

  import torch
  from bayesnf_amd import distributed
  from bayesnf_amd.engine import Engine
  from bayesnf_amd.spec import NetSpec

  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world > 1:
    distributed.maybe_init_from_env()
  rank = distributed.rank()
  if args.gpus != world and rank == 0:
    print(f'[bench] note: --gpus {args.gpus} but WORLD_SIZE {world}; using {world}',
          file=sys.stderr)

  X, y, input_scales = synthetic_grid()
  net = NetSpec(input_scales=input_scales, **MODEL_KW)
  E = args.members_per_gpu
  eng = Engine(net, mode='map', X=X, y=y, members=E, member_offset=rank * E, seed=0,
               learning_rate=0.005, prior_weight=1.0, compute_dtype=args.dtype)
  eng.init_params(float(np.log(np.nanstd(y) / 2)))

  def sync():
    torch.cuda.synchronize(eng.device)
    if world > 1:
      torch.distributed.barrier()
      torch.cuda.synchronize(eng.device)

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
  if world > 1:
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(tmax.item())
  final_loss = float(losses[:, -1].mean().item())
  if not np.isfinite(final_loss):
    raise RuntimeError('non-finite training loss in the benchmark run')

  if rank == 0:
    total_members = E * world
    value = total_members * args.steps / elapsed
    d = prof[dominant]
    achieved = d['flops'] / (d['avg_ms'] * 1e-3) / 1e12 if d['flops'] else 0.0
    peak = PEAK_TFLOPS[args.dtype]
    flops_step = net.flops_per_member_step(len(y)) * total_members
    line = {
        'metric': 'train-steps/sec x ensemble_size (member-steps/s, whole job)',
        'value': value, 'unit': 'member-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'C2 chickenpox-like MAP: N=10232 rows, D=3, F=57, width=512, '
                               'depth=2, NORMAL, full batch, lr=0.005',
                   'members_per_gpu': E, 'ensemble_size': total_members, 'parallelism':
                   f'ensemble-shard x{world} (no data-path collective)'},
        'algorithmic_tflops': flops_step * args.steps / elapsed / 1e12,
        'final_loss_mean': final_loss,
        'roofline': {'bound': 'mfma', 'kernel': dominant, 'achieved': achieved, 'peak': peak,
                     'unit': 'TFLOP/s', 'frac': achieved / peak,
                     'traffic': pmc_traffic(dominant, args.dtype, E),
                     'avg_launch_us': d['avg_ms'] * 1e3, 'launches': d['calls'],
                     'flops_per_launch': d['flops']},
    }
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(X, y, input_scales)
    print(json.dumps(line), flush=True)
  eng.close()
  if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
