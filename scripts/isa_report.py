#!/usr/bin/env python
"""Static report from hipcc's -save-temps assembly: per panel-kernel instantiation the register / scratch numbers, and
for one instantiation the instruction counts by class between workgroup barriers (static: loop bodies count once).

  hipcc ... -save-temps=obj bnf_api.hip   ->   python scripts/isa_report.py <file.s> ['<8, 4, true, false, 1, 64>']
"""
import re
import subprocess
import sys


def demangle(names):
  out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
  return out.splitlines()


def kind(op):
  if op.startswith('v_mfma'): return 'mfma'
  if op in ('v_exp_f32_e32', 'v_rcp_f32_e32', 'v_log_f32_e32', 'v_sqrt_f32_e32', 'v_rsq_f32_e32', 'v_sin_f32_e32', 'v_cos_f32_e32'): return 'trans'
  if op.startswith('v_'): return 'valu'
  if op.startswith('ds_'): return 'lds'
  if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
  if op == 's_nop': return 'nop'
  if op.startswith('s_'): return 'salu'
  return None


def main(path, want=None):
  lines = open(path).read().split('\n')
  starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r'^(_ZN3bnf\w+):', l)] if m]
  names = demangle([n for _, n in starts])
  sel = []
  print('| kernel | VGPRs | scratch | code bytes |')
  print('|---|---|---|---|')
  for k, (i, _) in enumerate(starts):
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    body = lines[i:end]
    def g(key):
      for l in body:
        m = re.match(r'^; ' + key + r'\s*[:=]\s*(\S+)', l.strip())
        if m:
          return m.group(1)
      return '?'
    nm = names[k].replace('void bnf::', '').split('(')[0]
    if 'panel' in nm or 'gemm_tn' in nm:
      print(f'| `{nm}` | {g("NumVgprs")} | {g("ScratchSize")} | {g("codeLenInByte")} |')
    if want and want in names[k] and 'k_panel_fwd_bwd' in names[k]:
      regions, cur = [], dict(valu=0, trans=0, lds=0, mfma=0, vmem=0, salu=0, nop=0)
      for l in body:
        m = re.match(r'^\s+([a-z_0-9]+)', l)
        if not m:
          continue
        if m.group(1) == 's_barrier':
          regions.append(cur)
          cur = dict(valu=0, trans=0, lds=0, mfma=0, vmem=0, salu=0, nop=0)
          continue
        k2 = kind(m.group(1))
        if k2:
          cur[k2] += 1
      regions.append(cur)
      sel = regions
  if want:
    print(f'\nstatic instruction counts between barriers, `{want}`:')
    for r in (sel if want else []):
      print(' ', r)


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
