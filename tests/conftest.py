"""pytest configuration: `gpu` marker (skipped, not failed, where no MI355X is visible) + repo
root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def _gpu_skip_reason():
  """None when the gpu tests can run here."""
  if not os.path.exists(os.path.join(ROOT, 'bayesnf_amd', 'libbnf_hip.so')):
    return 'bayesnf_amd/libbnf_hip.so is not built (python -c "import __graft_entry__ as g; g.build()")'
  try:
    import torch
    if not torch.cuda.is_available():
      return 'no GPU visible (torch.cuda.is_available() is False)'
  except Exception as e:   # pylint: disable=broad-except
    return f'torch unavailable: {e}'
  return None


def pytest_collection_modifyitems(config, items):
  """A plain `pytest tests/` on a CPU-only machine skips the gpu-marked tests instead of failing
  them.  On a GPU box nothing is skipped, and an explicit `-m gpu` without a GPU still fails
  loudly (the product has no CPU fallback; a silent all-skipped GPU tier would hide that)."""
  if 'gpu' in (config.getoption('-m') or '') and 'not gpu' not in (config.getoption('-m') or ''):
    return
  reason = _gpu_skip_reason()
  if reason is None:
    return
  skip = pytest.mark.skip(reason=reason)
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
  return GOLDEN
