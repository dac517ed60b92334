#!/usr/bin/env python
"""Copy the counter files a GPU visit produced into profiles/ and stamp the commit they belong to.

  python scripts/stamp_profiles.py gpurun_out/<tag>_pmc/traffic.json gpurun_out/<tag>/sq_counters.json

The visit wrote `_meta.kernel_source_sha16` (the GPU box has no .git); here, where the history is, `_meta.commit` becomes
the newest commit whose bayesnf_amd/csrc matches that hash -- i.e. HEAD when nothing under csrc/ changed since the visit
(otherwise the file is stale and bench.py will not quote it; this script says so)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main(traffic, counters):
  here = bench.kernel_source_sha16()
  head = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, text=True).strip()
  dirty = subprocess.check_output(['git', 'status', '--porcelain', 'bayesnf_amd/csrc'], cwd=ROOT, text=True).strip()
  for src, dst in ((traffic, 'pmc_traffic.json'), (counters, 'sq_counters.json')):
    with open(src) as f:
      table = json.load(f)
    meta = table.setdefault('_meta', {})
    ok = meta.get('kernel_source_sha16') == here
    meta['commit'] = (head + ('+csrc-uncommitted' if dirty else '')) if ok else 'unknown (kernel sources changed since the visit)'
    with open(os.path.join(ROOT, 'profiles', dst), 'w') as f:
      json.dump(table, f, indent=1)
    print(dst, 'matches this tree' if ok else 'STALE: taken with other kernel sources', meta)


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2])
