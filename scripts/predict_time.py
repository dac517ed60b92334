import time, numpy as np, torch
import sys; sys.path.insert(0, '.')
from bench import synthetic_grid, MODEL_KW
from bayesnf_amd.spec import NetSpec
from bayesnf_amd import inference
X, y, scales = synthetic_grid()
margs = dict(MODEL_KW, input_scales=scales)
import bayesnf_amd.inference as I
t0=time.perf_counter()
params, losses = I.fit_map(X, y, 0, 'NORMAL', margs, num_particles=64, learning_rate=0.005, num_epochs=50, compute_dtype='bf16')
torch.cuda.synchronize(); t1=time.perf_counter()
print('fit 50 epochs x 64 members: %.3f s' % (t1-t0))
for rep in range(2):
  t1=time.perf_counter()
  means, qs = I.predict_bnf(X, 'NORMAL', params, margs, (0.5, 0.025, 0.975), compute_dtype='bf16')
  t2=time.perf_counter()
  print('predict 64 members x %d rows, 3 root quantiles: %.3f s' % (X.shape[0], t2-t1), means.shape)
t2=time.perf_counter()
means, qs = I.predict_bnf(X, 'NORMAL', params, margs, (0.5, 0.025, 0.975), approximate_quantiles=True, compute_dtype='bf16')
print('predict approx quantiles: %.3f s' % (time.perf_counter()-t2))
