#!/usr/bin/env python
"""Per-kernel HBM traffic from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.

Units and gfx950 correction (MI355X_MICROARCH.md, HBM section): both counters are in
KiB; FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming reads on
gfx950, so reads are counted as 2 x FETCH_SIZE; WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def profile_meta():
  """What build the passes belong to (bench.py quotes the file only for a build with the same kernel sources):
  sha256 over bayesnf_amd/csrc (bench.kernel_source_sha16) + the date; `commit` is stamped when the file is copied
  into profiles/ (scripts/stamp_profiles.py: the GPU box has no .git)."""
  import datetime
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import bench
  return {'kernel_source_sha16': bench.kernel_source_sha16(), 'taken': datetime.date.today().isoformat(),
          'command': 'python bench.py --steps 10 --warmup 2 --no-cpu-baseline under rocprofv3 --pmc (one pass per counter set)'}


def load(counter_dir, counter):
  per = defaultdict(list)
  for path in glob.glob(os.path.join(counter_dir, '**', '*counter_collection*.csv'), recursive=True):
    with open(path) as f:
      for row in csv.DictReader(f):
        if row.get('Counter_Name') == counter:
          per[row['Kernel_Name']].append(float(row['Counter_Value']))
  return per


def main(out_dir):
  fetch = load(os.path.join(out_dir, 'FETCH_SIZE'), 'FETCH_SIZE')
  write = load(os.path.join(out_dir, 'WRITE_SIZE'), 'WRITE_SIZE')
  names = sorted(set(fetch) | set(write), key=lambda k: -(sum(fetch.get(k, [0])) + sum(write.get(k, [0]))))
  print('| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM bytes/launch (2*F + W) |')
  print('|---|---|---|---|---|')
  result = {}
  for k in names:
    f = fetch.get(k, [])
    w = write.get(k, [])
    fa = sum(f) / len(f) if f else 0.0
    wa = sum(w) / len(w) if w else 0.0
    traffic = (2 * fa + wa) * 1024
    result[k] = dict(launches=max(len(f), len(w)), fetch_kib=fa, write_kib=wa, hbm_bytes=traffic)
    short = k if len(k) < 90 else k[:87] + '...'
    print(f'| `{short}` | {max(len(f), len(w))} | {fa:.0f} | {wa:.0f} | {traffic:.3e} |')
  result['_meta'] = profile_meta()
  with open(os.path.join(out_dir, 'traffic.json'), 'w') as fjs:
    json.dump(result, fjs, indent=1)


if __name__ == '__main__':
  main(sys.argv[1])
