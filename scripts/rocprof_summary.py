#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table.

    python scripts/rocprof_summary.py gpurun_out/<tag>/prof/bench_results.db > profiles/<name>.md
"""
import sqlite3
import sys


def main(path, skip_first=0):
  db = sqlite3.connect(path)
  rows = db.execute(
      'select name, count(*), sum(duration), avg(duration), min(duration), max(duration), '
      'max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) '
      'from kernels group by name order by sum(duration) desc').fetchall()
  total = sum(r[2] for r in rows) or 1
  print(f'# rocprofv3 --kernel-trace summary of `{path}`\n')
  print('| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B | grid_x | wg |')
  print('|---|---|---|---|---|---|---|---|---|---|---|---|')
  for r in rows:
    name = r[0] if len(r[0]) < 110 else r[0][:107] + '...'
    print(f'| `{name}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | '
          f'{100*r[2]/total:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |')
  print(f'\ntotal kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches')


if __name__ == '__main__':
  main(sys.argv[1])
