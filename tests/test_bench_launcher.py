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


def test_inproc_launcher_one_process_drives_every_device():
  """`--launcher inproc` (the reference's own shape: one process, jax.pmap over the local devices): one host thread
  per device through `distributed.run_shards`, the timed region bracketed by a thread rendezvous + device sync on both
  sides, max over the devices, and the posterior gathered by ONE grouped all-gather over the local communicator
  set (`_native.allgather_local` -> bnf_comm_create_local / bnf_allgather_group; here bench.py's stand-in)."""
  r = _run(['--selftest-cpu', '--gpus', '3', '--launcher', 'inproc', '--steps', '3', '--warmup', '1', '--members-per-gpu', '2'])
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, r.stdout
  d = json.loads(lines[0])
  assert d['n_gpus'] == 3 and d['launcher'].startswith('inproc') and d['config']['ensemble_size'] == 6
  assert len(d['per_rank_ms_per_step']) == 3 and len(d['rank_devices']) == 3
  assert abs(d['ms_per_step'] - max(d['per_rank_ms_per_step'])) < 1e-9
  g = d['posterior_gather']
  assert 'bnf_allgather_group' in g['impl'] and g['shape'] == [6, 16] and g['rank_blocks_ok'] and g['rank_checksums_ok']
  assert g['cabi_calls'] == [['create_local', 3, [0, 1, 2]], ['allgather_group', 3, 2 * 16 * 4]]


def test_preflight_check_creates_the_communicator_and_exits():
  """`--check`: communicator(s) + a 1 KiB all-gather per rank, one JSON line, exit code 0 -- for both launchers; no
  engine, no data, seconds."""
  for extra in (['--launcher', 'inproc'], ['--gather', 'cabi'], []):
    r = _run(['--selftest-cpu', '--gpus', '2', '--check'] + extra)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['check'] == 'ok' and d['n_gpus'] == 2 and d['bytes_per_rank'] == 1024 and d['error'] is None
    assert d['launcher'] == ('inproc' if extra[:1] == ['--launcher'] else 'torchrun')


def test_committed_counter_files_are_quoted_only_for_the_build_they_measured(tmp_path, monkeypatch):
  """roofline.traffic / issue_floors come from committed rocprofv3 --pmc passes (they cannot be re-taken inside a
  timed run): the line says so (`traffic_source`), and a file taken with other kernel sources yields null."""
  import bench
  sha = bench.kernel_source_sha16()
  sym = 'void bnf::k_panel_fwd_bwd<8, 4, true, false, 1, 64>(bnf::PanelArgs)'
  prof = tmp_path / 'profiles'
  prof.mkdir()
  monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
  monkeypatch.setattr(bench, 'kernel_source_sha16', lambda: sha)
  (prof / 'pmc_traffic.json').write_text(json.dumps({sym: {'hbm_bytes': 2.5e9}, '_meta': {'kernel_source_sha16': sha, 'commit': 'abc1234'}}))
  v, src = bench.pmc_traffic('panel_fwd_bwd', 'bf16', 64)
  assert v == 2.5e9 and src['used'] and src['commit'] == 'abc1234' and src['file'].endswith('pmc_traffic.json')
  (prof / 'pmc_traffic.json').write_text(json.dumps({sym: {'hbm_bytes': 2.5e9}, '_meta': {'kernel_source_sha16': 'other'}}))
  v, src = bench.pmc_traffic('panel_fwd_bwd', 'bf16', 64)
  assert v is None and not src['used'] and 'stale' in src['why']
  (prof / 'pmc_traffic.json').write_text(json.dumps({sym: {'hbm_bytes': 2.5e9}}))          # no _meta at all: stale
  assert bench.pmc_traffic('panel_fwd_bwd', 'bf16', 64)[0] is None
  assert bench.pmc_traffic('panel_fwd_bwd', 'bf16', 8)[0] is None                           # other member count
  (prof / 'sq_counters.json').write_text(json.dumps({sym: {'SQ_INSTS_VALU': 2.0e8, 'SQ_INSTS_MFMA': 2.5e7},
                                                      '_meta': {'kernel_source_sha16': sha}}))
  ctr, src = bench.sq_counters('panel_fwd_bwd', 'bf16', 64)
  fl = bench.issue_floors(ctr, 1200.0)
  assert src['used'] and abs(fl['valu_per_mfma'] - 8.0) < 1e-9 and fl['valu_issue_floor_us'] > 0


def test_inproc_and_torchrun_launchers_split_a_strong_scaling_ensemble_the_same_way():
  """`--strong`: `--members-per-gpu` names the WHOLE ensemble and both launchers must deal it out identically -- same
  ensemble size, same members per GPU, same scaling label, same gather shape -- so that a strong-scaling record taken
  with one launcher is comparable with the other's."""
  common = ['--selftest-cpu', '--gpus', '2', '--steps', '2', '--warmup', '1', '--members-per-gpu', '6', '--strong']
  out = {}
  for name, extra in (('torchrun', []), ('inproc', ['--launcher', 'inproc'])):
    r = _run(common + extra)
    assert r.returncode == 0, r.stderr[-2000:]
    out[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
  a, b = out['torchrun'], out['inproc']
  for d in (a, b):
    assert d['scaling'] == 'strong' and d['n_gpus'] == 2
    assert d['config']['ensemble_size'] == 6 and d['config']['members_per_gpu'] == 3
    assert len(d['per_rank_ms_per_step']) == 2 and d['posterior_gather']['shape'] == [6, 16]
    assert abs(d['value'] - 6 * 2 / (d['ms_per_step'] * 2e-3)) < 1e-6 * d['value']      # whole-job member-steps/s
  assert a['launcher'].startswith('torch') and b['launcher'].startswith('inproc')
  assert a['metric'] == b['metric'] and a['unit'] == b['unit'] and a['config']['parallelism'] == b['config']['parallelism']
  # an ensemble that does not split evenly is refused by both
  for extra in ([], ['--launcher', 'inproc']):
    r = _run(['--selftest-cpu', '--gpus', '2', '--steps', '1', '--warmup', '1', '--members-per-gpu', '5', '--strong'] + extra)
    assert r.returncode != 0 and 'do not split evenly' in (r.stderr + r.stdout)
