import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_diag import stage_diag
import numpy as np
for (d, w, n) in [(1, 128, 130), (2, 128, 300), (2, 256, 300), (2, 512, 200)]:
  for pl in ('layers', 'fused'):
    print(f'=== bf16 depth{d} W{w} {pl}')
    stage_diag('bf16', d, w, n, pipeline=pl)
