"""What would fp8 FORWARD and BACKWARD-DATA contractions cost in accuracy?  (VERDICT r05 item 3 / SURVEY row R1: BASELINE
configs[4] says "fp8 MFMA dense layers"; the engine's compute_dtype='fp8' keeps those contractions bf16 and stores only the
weight-gradient operands as fp8.)  CPU emulation on the torch port of the oracle (oracle/torch_baseline.py: hand-derived
backward), with the operands of every Dense contraction rounded the way the kernels would:

  fp32    : nothing rounded
  bf16    : H_l, K_l (forward), dZ_l, K_l (backward-data), H_l, dZ_l (weight gradient) rounded to bf16          = the bf16 engine
  fp8w    : bf16, but the weight-gradient operands H_l -> OCP e4m3, dZ_l -> OCP e5m2 / s_dZ                     = today's 'fp8'
  fp8all  : + forward H_l e4m3 x K_l e4m3 (K scaled by 2^5 per layer), backward-data dZ_l e5m2 / s_dZ x K_l e4m3 = R1 complete
  fp8ww   : fp8all on the W x W layers only (layer 0 forward / backward-data stay bf16)

and SURVEY 8d's fp8 gate measured for each: 150 full-batch Adam steps from identical initial parameters at C2's and C5's
feature layouts and widths, final loss within 3 % and RMSE of the ensemble-mean prediction within 5 % of the fp32 run.
Accumulation is f32 everywhere (as on the MFMA).  s_dZ = 2^(round(log2(c gamma_o / sigma)) - 6) per member, like the kernel.
Usage: python scripts/fp8_forward_emulation.py [C2|C5] [steps] [rows]"""
import math
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from oracle import bnf_oracle as O            # noqa: E402
from oracle.torch_baseline import TorchStep   # noqa: E402
from tests import util                        # noqa: E402


def q_bf16(x):
  return x.to(torch.bfloat16).float()


def q_e4m3(x, scale=1.0):
  return (x * scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() / scale


def q_e5m2(x, scale):          # scale: (E, 1, 1) powers of two; stored value = x / scale
  return (x / scale).clamp(-57344.0, 57344.0).to(torch.float8_e5m2).float() * scale


class QuantStep(TorchStep):
  """TorchStep with operand rounding at the three contractions of every hidden Dense layer."""

  def __init__(self, *a, mode='fp32', **kw):
    super().__init__(*a, **kw)
    self.mode = mode

  def _ops(self, l):
    """(forward H, forward K, bwd-data dZ, bwd-data K, wgrad H, wgrad dZ) rounding functions of layer l"""
    ident = lambda x, *_: x
    m = self.mode
    if m == 'fp32':
      return (ident,) * 6
    b = lambda x, *_: q_bf16(x)
    if m == 'bf16':
      return (b,) * 6
    h8 = lambda x, *_: q_e4m3(x)
    k8 = lambda x, *_: q_e4m3(x, 32.0)
    z8 = lambda x, s: q_e5m2(x, s)
    if m == 'fp8w':
      return (b, b, b, b, h8, z8)
    if m == 'fp8all' or (m == 'fp8ww' and l >= 1):
      return (h8, k8, z8, k8, h8, z8)
    if m == 'fp8ww':
      return (b, b, b, b, h8, z8)
    raise ValueError(m)

  def loss_and_grad(self, theta):
    m, X, y = self.m, self.X, self.y
    E, B = theta.shape[0], X.shape[0]
    W, L = m.width, m.depth
    c = self.n_total / B
    g = torch.zeros_like(theta)

    def put(name, val):
      lf = m.leaf[name]
      g[:, lf.offset:lf.offset + lf.size] += val.reshape(E, lf.size)

    lsa = self._v(theta, 'log_scale_adjustment')
    s = torch.as_tensor(m.input_scales.astype(np.float32)) * torch.exp(lsa)
    u = X[None] / s[:, None, :]
    G, fargs = [], {}
    for kind, arg, ncols, col0, sname in m.groups:
      if kind == 'u':
        G.append(u)
      elif kind == 'fourier':
        deg = ncols // 2
        a = self.fconst[arg] * u[..., arg, None]
        fargs[arg] = a
        den = torch.arange(1, deg + 1, dtype=torch.float32)
        G.append(torch.cat([torch.cos(a) / den, torch.sin(a) / den], dim=-1))
      elif kind == 'seasonal':
        G.append(self.seas[None].expand(E, -1, -1))
      else:
        p, q = m.interactions[:, 0], m.interactions[:, 1]
        G.append(u[..., p] * u[..., q])
    H0 = torch.cat([gg * F.softplus(self._v(theta, grp[4]))[:, None, None] for gg, grp in zip(G, m.groups)], dim=-1)
    alpha = torch.sigmoid(self._v(theta, 'logit_activation_weight'))[:, None, None]
    Hs, As, gams = [H0], [], []
    h = H0
    for l in range(L):
      n = h.shape[-1]
      fH, fK = self._ops(l)[:2]
      K = self._v(theta, f'Dense_{l}/kernel')
      b = self._v(theta, f'Dense_{l}/bias')
      gam = F.softplus(self._v(theta, f'inv_sp_layer_scale{l}'))[:, None, None]
      a = gam * (torch.baddbmm(b[:, None, :], fH(h), fK(K), alpha=1.0 / math.sqrt(n)))
      th_, el = torch.tanh(a), F.elu(a)
      h = th_ + alpha * (el - th_)
      As.append((a, th_, el))
      gams.append(gam)
      Hs.append(h)
    ko = self._v(theta, f'Dense_{L}/kernel')[..., 0]
    bo = self._v(theta, f'Dense_{L}/bias')[..., 0]
    gam_o = F.softplus(self._v(theta, 'inv_sp_output_scale'))
    v = torch.bmm(h, ko[:, :, None])[..., 0] / math.sqrt(W) + bo[:, None]
    out = gam_o[:, None] * v
    lns = self._v(theta, 'log_noise_scale')
    sigma = 0.01 + torch.exp(lns)
    res = y[None] - out
    z = res / sigma[:, None]
    ll = torch.sum(-0.5 * z * z - torch.log(sigma)[:, None] - 0.5 * math.log(2 * math.pi), dim=-1)
    loss = -c * ll
    s_dz = torch.exp2(torch.round(torch.log2(c * gam_o / sigma)) - 6.0)[:, None, None]
    put('log_noise_scale', -c * torch.sum(res * res / sigma[:, None]**3 - 1.0 / sigma[:, None], dim=-1) * torch.exp(lns))
    dout = -c * res / sigma[:, None]**2
    put('inv_sp_output_scale', torch.sigmoid(self._v(theta, 'inv_sp_output_scale')) * torch.sum(dout * v, dim=-1))
    dv = gam_o[:, None] * dout
    sW = math.sqrt(W)
    put(f'Dense_{L}/kernel', torch.bmm(Hs[L].transpose(1, 2), dv[:, :, None])[..., 0] / sW)
    put(f'Dense_{L}/bias', dv.sum(dim=-1))
    dH = dv[:, :, None] * ko[:, None, :] / sW
    dalpha = torch.zeros(E)
    for l in range(L - 1, -1, -1):
      _, _, fZd, fKd, fHw, fZw = self._ops(l)
      a, th_, el = As[l]
      gam = gams[l]
      dalpha += torch.sum(dH * (el - th_), dim=(1, 2))
      dact = alpha * torch.where(a > 0, torch.ones(()), torch.exp(torch.clamp(a, max=0.0))) + (1 - alpha) * (1 - th_ * th_)
      dA = dH * dact
      put(f'inv_sp_layer_scale{l}', torch.sigmoid(self._v(theta, f'inv_sp_layer_scale{l}')) *
          torch.sum(dA * (a / gam), dim=(1, 2)))
      dZ = gam * dA
      Hl = Hs[l]
      sn = math.sqrt(Hl.shape[-1])
      put(f'Dense_{l}/kernel', torch.bmm(fHw(Hl).transpose(1, 2), fZw(dZ, s_dz)) / sn)
      put(f'Dense_{l}/bias', dZ.sum(dim=1))
      dH = torch.bmm(fZd(dZ, s_dz), fKd(self._v(theta, f'Dense_{l}/kernel')).transpose(1, 2)) / sn
    a1 = alpha[:, 0, 0]
    put('logit_activation_weight', a1 * (1 - a1) * dalpha)
    du = torch.zeros_like(u)
    for gidx, (kind, arg, ncols, col0, sname) in enumerate(m.groups):
      dHg = dH[..., col0:col0 + ncols]
      fs = self._v(theta, sname)
      put(sname, torch.sigmoid(fs) * torch.sum(dHg * G[gidx], dim=(1, 2)))
      dG = F.softplus(fs)[:, None, None] * dHg
      if kind == 'u':
        du += dG
      elif kind == 'fourier':
        deg = ncols // 2
        a = fargs[arg]
        den = torch.arange(1, deg + 1, dtype=torch.float32)
        du[..., arg] += torch.sum(self.fconst[arg] * (-torch.sin(a) * dG[..., :deg] + torch.cos(a) * dG[..., deg:]) / den,
                                  dim=-1)
      elif kind == 'inter':
        for k, (p, q) in enumerate(m.interactions):
          du[..., p] += dG[..., k] * u[..., q]
          du[..., q] += dG[..., k] * u[..., p]
    put('log_scale_adjustment', -torch.sum(du * u, dim=1))
    if self.pw != 0.0:
      zt = theta - self.prior_loc
      loss = loss - self.pw * torch.sum(-zt - 2.0 * F.softplus(-zt), dim=-1)
      g += self.pw * torch.tanh(0.5 * zt)
    return loss, g


def main():
  layout = sys.argv[1] if len(sys.argv) > 1 else 'C2'
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
  if layout == 'C2':
    kw = dict(n_rows=4000, width=512, depth=2, periods=(4.0, 52.1775), harmonics=(2, 10), T=522)
  else:
    kw = dict(n_rows=6000, width=256, depth=2, periods=(7.0, 30.4375, 365.25), harmonics=(3, 10, 10), T=2000, interactions=())
  if len(sys.argv) > 3:
    kw['n_rows'] = int(sys.argv[3])
  net, model, X, y = util.make_problem(**kw)
  E = 8
  rng = np.random.default_rng(3)
  theta0 = O.map_init(model, y, rng.standard_normal((E, model.P)).clip(-2, 2), dtype=np.float32)
  print(f'{layout}: rows {kw["n_rows"]} W {kw["width"]} F {model.F} members {E} steps {steps} (the gate: |loss / loss_fp32 - 1| < 0.03, '
        f'|rmse / rmse_fp32 - 1| < 0.05)', flush=True)
  ref = None
  for mode in ('fp32', 'bf16', 'fp8w', 'fp8ww', 'fp8all'):
    t0 = time.time()
    st = QuantStep(model, X, y, prior_weight=1.0, lr=0.005, mode=mode)
    th, losses = st.train(theta0, steps)
    pred = np.asarray(O.forward(model, th.astype(np.float64), X)).mean(axis=0)
    rmse = float(np.sqrt(np.mean((pred - y) ** 2)))
    lf = float(np.mean(losses[:, -1]))
    if ref is None:
      ref = (lf, rmse)
    print(f'  {mode:7s}: final loss {lf:12.3f} ({lf / ref[0] - 1:+.4f})   rmse {rmse:.5f} ({rmse / ref[1] - 1:+.4f})   '
          f'[{time.time() - t0:.0f} s]', flush=True)


if __name__ == '__main__':
  main()
