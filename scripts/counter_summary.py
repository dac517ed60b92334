#!/usr/bin/env python
"""Per-kernel averages (per launch) of whatever counters the passes under <dir>/pass*/ hold."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(out_dir):
  vals = defaultdict(lambda: defaultdict(list))
  for path in glob.glob(os.path.join(out_dir, 'pass*', '**', '*counter_collection*.csv'), recursive=True):
    with open(path) as f:
      for row in csv.DictReader(f):
        vals[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
  counters = sorted({c for k in vals for c in vals[k]})
  kernels = [k for k in vals if k.startswith(('void bnf', 'bnf::'))]
  key = 'SQ_BUSY_CYCLES' if 'SQ_BUSY_CYCLES' in counters else (counters[0] if counters else None)
  kernels.sort(key=lambda k: -sum(vals[k].get(key, [0])))
  short = lambda k: (k.replace('void ', '').replace('bnf::', '').split('(')[0])[:34]
  print('| counter (avg / launch) | ' + ' | '.join(short(k) for k in kernels) + ' |')
  print('|---|' + '---|' * len(kernels))
  for c in counters:
    cells = []
    for k in kernels:
      v = vals[k].get(c)
      cells.append(f'{sum(v) / len(v):.4g}' if v else '-')
    print(f'| {c} | ' + ' | '.join(cells) + ' |')


def dump_json(out_dir, path):
  """per-kernel per-launch averages as JSON (profiles/sq_counters.json, read by bench.py's roofline block)"""
  import json
  vals = defaultdict(lambda: defaultdict(list))
  for f in glob.glob(os.path.join(out_dir, 'pass*', '**', '*counter_collection*.csv'), recursive=True):
    with open(f) as fh:
      for row in csv.DictReader(fh):
        vals[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
  out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in vals.items() if k.startswith(('void bnf', 'bnf::'))}
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from pmc_summary import profile_meta
  out['_meta'] = profile_meta()
  with open(path, 'w') as fh:
    json.dump(out, fh, indent=1)


if __name__ == '__main__':
  main(sys.argv[1])
  if len(sys.argv) > 2:
    dump_json(sys.argv[1], sys.argv[2])
