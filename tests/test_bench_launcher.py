"""bench.py's own multi-rank path on the CPU: `python bench.py --gpus 2` starts two ranks by itself
(torch.distributed.run on 127.0.0.1), proves the collective saw both (all_reduce), reduces the
timing as max over ranks, gathers the per-rank predictive means with ONE all_gather_into_tensor and
prints ONE JSON line.  `--selftest-cpu` swaps the engine for a stand-in step and RCCL for gloo; every
other line of that path is the one the GPU run executes.  Also: asking for more GPUs than are
visible fails loudly instead of measuring a smaller job."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=300):
  env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
  env['OMP_NUM_THREADS'] = '2'
  return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True,
                        timeout=timeout, env=env, cwd=ROOT)


def test_two_gloo_ranks_end_to_end_through_bench():
  r = _run(['--selftest-cpu', '--gpus', '2', '--steps', '4', '--warmup', '1', '--members-per-gpu', '3'])
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, r.stdout            # exactly one JSON line, from rank 0
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['rccl_world_size'] == 2 and d['selftest'] is True
  assert d['config']['ensemble_size'] == 6 and d['scaling'] == 'weak'
  assert len(d['per_rank_ms_per_step']) == 2
  assert abs(d['ms_per_step'] - max(d['per_rank_ms_per_step'])) < 1e-9      # max over ranks
  assert abs(d['value'] - 6 * 4 / (d['ms_per_step'] * 4e-3)) < 1e-6 * d['value']
  g = d['posterior_gather']
  assert g['shape'] == [6, 16] and g['rank_blocks_ok'] and g['finite'] and 'all_gather_into_tensor' in g['impl']


def test_cabi_gather_plumbing_and_strong_scaling_flag():
  """`--gather cabi` (the GPU default): `_native.allgather` makes the communicator id on rank 0, carries it
  to the other rank over the process group, creates one communicator per rank and calls bnf_allgather with
  the per-rank byte count -- here against bench.py's stand-in for the three C entry points (no GPU); the
  per-rank checksums gathered beside the payload match the delivered blocks.  `--strong`: the ensemble
  size is fixed and split over the ranks."""
  r = _run(['--selftest-cpu', '--gpus', '2', '--steps', '2', '--warmup', '1', '--members-per-gpu', '6', '--strong',
            '--gather', 'cabi'])
  assert r.returncode == 0, r.stderr[-2000:]
  d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
  assert d['scaling'] == 'strong' and d['config']['ensemble_size'] == 6 and d['config']['members_per_gpu'] == 3
  g = d['posterior_gather']
  assert 'bnf_allgather' in g['impl'] and g['shape'] == [6, 16] and g['rank_blocks_ok'] and g['rank_checksums_ok']
  assert g['rank_checksums'] == [0.0, 48.0]                      # rank r contributes 3 x 16 values r
  assert g['cabi_calls'][0] == ['create', 2, 0, 0]               # rank 0's view: world 2, rank 0
  assert ['allgather', 3 * 16 * 4] in g['cabi_calls'] and g['cabi_id_head'] == [3, 10, 17, 24]
  r = _run(['--selftest-cpu', '--gpus', '2', '--steps', '1', '--warmup', '1', '--members-per-gpu', '5', '--strong'])
  assert r.returncode != 0 and 'do not split evenly' in (r.stderr + r.stdout)


def test_single_rank_line_has_the_contract_fields():
  r = _run(['--selftest-cpu', '--steps', '3', '--warmup', '1'])
  assert r.returncode == 0, r.stderr[-2000:]
  d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
  for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config'):
    assert key in d
  assert d['n_gpus'] == 1 and d['rccl_world_size'] == 1 and 'posterior_gather' not in d


def test_more_gpus_than_visible_fails_loudly():
  r = _run(['--gpus', '2', '--steps', '1', '--warmup', '1'])
  assert r.returncode != 0
  assert 'devices visible: 0' in (r.stderr + r.stdout)
