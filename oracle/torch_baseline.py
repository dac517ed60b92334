"""CPU baseline of the train step: the oracle's algorithm on torch CPU tensors.

THIS IS TEST / MEASUREMENT INFRASTRUCTURE (see oracle/bnf_oracle.py): only bench.py's
`cpu_baseline` leg and tests/ import it.  It is the SAME restatement as
`bnf_oracle.map_loss_and_grad` + `adam_update` (reference models.py:212-273, inference.py:558-569,
580-606; hand-derived backward) for the NORMAL observation model, written on
float32 torch tensors so that the host's cores are actually used: members are batched through
`torch.bmm` (oneDNN / MKL GEMM) and every element-wise pass is multi-threaded (numpy's are
single-threaded, which is what kept the numpy oracle at ~1 % of the host's GEMM rate).
tests/test_torch_baseline.py checks it against the numpy oracle (loss 1e-5, gradients 1e-4).
It is a port, not JAX: the JAX reference cannot be installed here (no network).
"""

from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import bnf_oracle as O


class TorchStep:
  """Full-batch (or fixed-batch) MAP / MLE step of E members on the CPU, float32."""

  def __init__(self, model: O.Model, X, y, n_total=None, prior_weight=1.0, lr=0.005):
    assert model.observation_model == 'NORMAL'
    self.m = model
    self.X = torch.as_tensor(np.asarray(X, dtype=np.float32))
    self.y = torch.as_tensor(np.asarray(y, dtype=np.float32))
    self.n_total = float(n_total if n_total is not None else len(y))
    self.pw, self.lr = float(prior_weight), float(lr)
    self.prior_loc = torch.as_tensor(model.prior_loc().astype(np.float32))
    # data-constant trig arguments in the reference's float32 semantics
    two_pi = np.float32(2.0 * np.pi)
    t = self.X[:, 0]
    self.sarg = torch.as_tensor((two_pi * model.freqs.astype(np.float32)).astype(np.float32))[None, :] * t[:, None]
    self.seas = torch.cat([torch.cos(self.sarg), torch.sin(self.sarg)], dim=-1) / torch.as_tensor(
        np.tile(model.harm.astype(np.float32), 2))
    self.fconst = {d: torch.as_tensor(two_pi * np.float32(2.0)**np.arange(deg, dtype=np.float32))
                   for d, deg in enumerate(model.fourier_degrees) if deg > 0}

  def _v(self, theta, name):
    lf = self.m.leaf[name]
    return theta[:, lf.offset:lf.offset + lf.size].reshape((theta.shape[0],) + tuple(lf.shape))

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
    u = X[None] / s[:, None, :]                                  # (E,B,D)
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
      K = self._v(theta, f'Dense_{l}/kernel')
      b = self._v(theta, f'Dense_{l}/bias')
      gam = F.softplus(self._v(theta, f'inv_sp_layer_scale{l}'))[:, None, None]
      a = gam * (torch.baddbmm(b[:, None, :], h, K, alpha=1.0 / math.sqrt(n)))
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
    # ---- backward
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
      put(f'Dense_{l}/kernel', torch.bmm(Hl.transpose(1, 2), dZ) / sn)
      put(f'Dense_{l}/bias', dZ.sum(dim=1))
      dH = torch.bmm(dZ, self._v(theta, f'Dense_{l}/kernel').transpose(1, 2)) / sn
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

  def train(self, theta0, steps):
    theta = torch.as_tensor(np.asarray(theta0, dtype=np.float32)).clone()
    mom, vel = torch.zeros_like(theta), torch.zeros_like(theta)
    losses = []
    with torch.no_grad():
      for t in range(1, steps + 1):
        loss, g = self.loss_and_grad(theta)
        mom.mul_(0.9).add_(g, alpha=0.1)
        vel.mul_(0.999).addcmul_(g, g, value=0.001)
        theta -= self.lr * (mom / (1 - 0.9**t)) / (torch.sqrt(vel / (1 - 0.999**t)) + 1e-8)
        losses.append(loss)
    return theta.numpy(), torch.stack(losses, dim=1).numpy()
