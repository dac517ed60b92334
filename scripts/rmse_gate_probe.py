import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests import test_gpu_fullsize as T
from bayesnf_amd.engine import Engine
X, y, scales = T._grid(); net = T._net(scales)
kw = dict(seed=13, learning_rate=0.005, members=8)
fwd = Engine(net, members=8, forward_only=True, row_capacity=4096, compute_dtype='fp32')
Xd = torch.tensor(X, dtype=torch.float32, device=fwd.device)
def rm(th):
  loc, _ = fwd.forward(torch.tensor(th, dtype=torch.float32, device=fwd.device), Xd); torch.cuda.synchronize()
  pred = loc.cpu().numpy()
  return np.sqrt(np.mean((pred - y[None, :]) ** 2, axis=1)), np.sqrt(np.mean((pred.mean(axis=0) - y) ** 2))
th32, l32 = T._fit(net, X, y, 200, compute_dtype='fp32', **kw); r32 = rm(th32)
worst = {}
for dt, n in (('bf16', 60), ('fp32', 15)):
  stats = []
  for i in range(n):
    th, l = T._fit(net, X, y, 200, compute_dtype=dt, **kw); r = rm(th)
    dev = r[0]/r32[0]-1
    if np.abs(dev).max() > 0.05: print('   outlier run', i, dt, 'member', int(np.abs(dev).argmax()), np.round(dev, 3))
    stats.append((np.abs(r[0]/r32[0]-1).max(), abs(r[0].mean()/r32[0].mean()-1), abs(r[1]/r32[1]-1), np.abs(l[:,-1]/l32[:,-1]-1).max(), float(np.isfinite(th).all())))
  s = np.array(stats)
  print(dt, n, 'runs: member max: median %.4f max %.4f | mean-of-members max %.4f | ensemble max %.4f | final loss max %.5f | finite %s' % (np.median(s[:,0]), s[:,0].max(), s[:,1].max(), s[:,2].max(), s[:,3].max(), s[:,4].min()))
  print('   member-max sorted tail', np.sort(s[:,0])[-5:])
