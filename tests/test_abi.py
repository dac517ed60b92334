"""CPU-only checks of the C-ABI shared library: it loads, exports every symbol
include/bnf.h declares, the ctypes mirror of `bnf_config` has the C layout, and
compute entry points refuse loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

from bayesnf_amd import _native
from bayesnf_amd.spec import NetSpec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'bnf.h')


def _declared():
  src = open(HEADER).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(bnf_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
  lib = _native.load()
  names = _declared()
  assert 'bnf_train' in names and 'bnf_forward' in names and len(names) >= 20
  for n in names:
    assert hasattr(lib, n), f'{n} declared in include/bnf.h but not exported'
  assert set(names) == set(_native.EXPORTS)
  assert lib.bnf_abi_version() == _native.ABI_VERSION


def test_config_struct_layout_matches_c(tmp_path):
  """sizeof/offsetof from a C compile of include/bnf.h vs the ctypes mirror."""
  probe = tmp_path / 'probe.c'
  fields = ['abi_version', 'n_groups', 'group_scale_off', 'input_scale', 'n_freqs', 'harmonic',
            'interact', 'off_bias', 'off_act_weight', 'n_rows', 'batch', 'members',
            'member_offset', 'vi_samples', 'forward_only', 'pipeline', 'learning_rate', 'kl_weight', 'seed']
  body = '\n'.join(f'  printf("{f} %zu\\n", offsetof(bnf_config, {f}));' for f in fields)
  probe.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "bnf.h"\nint main(){\n'
                   f'  printf("sizeof %zu\\n", sizeof(bnf_config));\n{body}\n  return 0;}}\n')
  exe = tmp_path / 'probe'
  subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(probe), '-o', str(exe)])
  out = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
  assert int(out['sizeof']) == C.sizeof(_native.BnfConfig)
  for f in fields:
    assert int(out[f]) == getattr(_native.BnfConfig, f).offset, f


def test_make_config_roundtrip():
  net = NetSpec(width=128, depth=2, input_scales=[51, 1, 1], fourier_degrees=[5, 5, 5],
                interactions=[(0, 1)], seasonality_periods=[4, 52.1775],
                num_seasonal_harmonics=[2, 10])
  c = _native.make_config(net, device=0, dtype='bf16', mode=_native.MODE_MAP, n_rows=1000,
                          batch=250, members=6, member_offset=12, seed=(1 << 40) + 5,
                          learning_rate=0.01, prior_weight=0.0)
  assert (c.n_features, c.n_params, c.n_groups) == (net.F, net.P, len(net.groups))
  assert c.dtype == 1 and c.batch == 250 and c.member_offset == 12 and c.seed == (1 << 40) + 5
  assert c.off_kernel[2] == net.offset('Dense_2/kernel')
  assert list(c.interact[0]) == [0, 1] and c.n_freqs == 12
  assert abs(c.freq[2] - 1 / 52.1775) < 1e-7
  # any width is accepted (the engine pads to the next multiple of 64 internally); 0 is not
  c100 = _native.make_config(NetSpec(width=100, depth=1, input_scales=[1], fourier_degrees=[1],
                                     interactions=[]), device=0, dtype='fp32', mode=0, n_rows=4,
                             batch=4, members=1, member_offset=0, seed=0)
  assert c100.width == 100
  with pytest.raises(ValueError):
    _native.make_config(NetSpec(width=0, depth=1, input_scales=[1], fourier_degrees=[1],
                                interactions=[]), device=0, dtype='fp32', mode=0, n_rows=4,
                        batch=4, members=1, member_offset=0, seed=0)


def test_seed_helpers():
  assert _native.seed_to_u64(7) == 7
  assert _native.seed_to_u64([1, 2]) == (1 << 32) | 2
  assert _native.fold_in(5, 0) != _native.fold_in(5, 1) != 5
  with pytest.raises(ValueError):
    _native.seed_to_u64([1, 2, 3])


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_no_cpu_fallback():
  lib = _native.load()
  net = NetSpec(width=64, depth=1, input_scales=[1], fourier_degrees=[1], interactions=[])
  cfg = _native.make_config(net, device=0, dtype='fp32', mode=0, n_rows=4, batch=4, members=1,
                            member_offset=0, seed=0)
  h = C.c_void_p()
  assert lib.bnf_create(C.byref(cfg), C.byref(h)) == -2
  assert 'no CPU fallback' in _native.last_error()
  from bayesnf_amd.engine import Engine
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    Engine(net, X=[[0.0]] * 4, y=[0.0] * 4)
  import pandas as pd
  from bayesnf_amd import BayesianNeuralFieldMAP
  df = pd.DataFrame({'t': pd.date_range('2020-01-06', periods=8, freq='W-MON'),
                     'y': range(8)})
  est = BayesianNeuralFieldMAP(feature_cols=['t'], target_col='y', freq='W', width=64)
  with pytest.raises(RuntimeError):
    est.fit(df, seed=0, ensemble_size=2, num_epochs=1)


def test_dtype_names_and_the_announced_default(monkeypatch):
  """An explicit 'fp32' is ALWAYS the exact f32 chain (BNF_DTYPE_F32 = 0); the split-bf16 form has its own name
  ('fp32_split' = 3) and is what the estimators run when nothing is said -- announced once per process."""
  import warnings
  from bayesnf_amd import engine
  assert [_native.DTYPE[n] for n in ('fp32', 'f32', 'float32', 'fp32_exact')] == [0, 0, 0, 0]
  assert [_native.DTYPE[n] for n in ('fp32_split', 'bf16x3', 'bf16', 'fp8')] == [3, 3, 1, 2]
  assert _native.ABI_VERSION == 6
  monkeypatch.delenv('BNF_DTYPE', raising=False)
  for name, canon in (('fp32', 'fp32'), ('fp32_exact', 'fp32'), ('fp32_split', 'fp32_split'), ('bf16', 'bf16'), ('fp8', 'fp8')):
    assert engine.default_dtype(name) == canon
  monkeypatch.setattr(engine, '_warned_default_dtype', False)
  with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    assert engine.default_dtype(None) == 'fp32_split'
    assert engine.default_dtype(None) == 'fp32_split'
  assert len(w) == 1 and "defaults to 'fp32_split'" in str(w[0].message) and "'fp32' for the exact" in str(w[0].message)
  monkeypatch.setenv('BNF_DTYPE', 'fp32')
  assert engine.default_dtype(None) == 'fp32' and engine.default_dtype('bf16') == 'bf16'
  import pytest
  with pytest.raises(ValueError):
    engine.default_dtype('fp16')
