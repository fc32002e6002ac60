"""CPU restatement of the flow-matching UniPC sampler step (TEST INFRASTRUCTURE).

Follows seaweed_apt/wan/utils/fm_solvers_unipc.py for the configuration
WanT2V.generate uses (text2video.py:205-211): order 2, bh2, predict_x0,
flow_prediction, lower_order_final, final sigma 0.  Written tensor-by-tensor
(not through the product's coefficient folding) so it is an independent check
of omnihuman-1-hack_amd/wan/utils/fm_solvers_unipc.py.

Pinned against the real reference class by oracle/make_golden.py (the class
imports with a ~20-line stub of diffusers' SchedulerMixin/ConfigMixin).
"""
import numpy as np
import torch


def sampling_sigmas(num_steps: int, shift: float, num_train_timesteps: int = 1000):
    """fm_solvers_unipc.py:106-131,160-208 — float32 sigmas [n+1] (last 0) and int64 timesteps [n]."""
    alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
    base = torch.from_numpy(1.0 - alphas).to(torch.float32)          # constructor shift = 1
    smax, smin = base[0].item(), base[-1].item()
    s = np.linspace(smax, smin, num_steps + 1).copy()[:-1]
    s = shift * s / (1 + (shift - 1) * s)
    timesteps = torch.from_numpy(s * num_train_timesteps).to(torch.int64)
    sig = torch.from_numpy(np.concatenate([s, [0]]).astype(np.float32))
    return sig, timesteps


class UniPCOracle:
    def __init__(self, num_steps: int, shift: float, solver_order: int = 2):
        self.sigmas, self.timesteps = sampling_sigmas(num_steps, shift)
        self.order = solver_order
        self.m = [None] * solver_order       # x0 predictions, oldest first
        self.lower = 0
        self.last = None
        self.this_order = None
        self.i = 0

    @staticmethod
    def _lam(s):
        return torch.log(1 - s) - torch.log(s)

    def _update(self, x, m0, older, s_t, s_0, s_older, order, model_t=None):
        """Shared predictor (model_t None, :404-470) / corrector (:548-619) algebra."""
        a_t = 1 - s_t
        h = self._lam(s_t) - self._lam(s_0)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        x_t_ = s_t / s_0 * x - a_t * h_phi_1 * m0
        res = 0
        rk = None
        if order == 2:
            rk = (self._lam(s_older) - self._lam(s_0)) / h
            D1 = (older - m0) / rk
        if model_t is None:                       # predictor
            if order == 2:
                res = 0.5 * D1
            return x_t_ - a_t * B_h * res
        if order == 1:
            rhos = torch.tensor([0.5])
        else:
            h_phi_k = h_phi_1 / hh - 1
            b, fact = [], 1
            for k in range(1, 3):
                b.append(h_phi_k * fact / B_h)
                fact *= k + 1
                h_phi_k = h_phi_k / hh - 1 / fact
            R = torch.stack([torch.ones(2), torch.stack([rk, torch.tensor(1.0)])])
            rhos = torch.linalg.solve(R, torch.stack(b))
            res = rhos[0] * D1
        return x_t_ - a_t * B_h * (res + rhos[-1] * (model_t - m0))

    def step(self, v: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """One scheduler.step(v, t_i, x) (:655-739)."""
        i, sig = self.i, self.sigmas
        m_t = x - sig[i] * v
        if i > 0 and self.last is not None:
            x = self._update(self.last, self.m[-1], self.m[-2] if self.order > 1 else None, sig[i], sig[i - 1],
                             sig[i - 2] if i >= 2 else None, self.this_order, model_t=m_t)
        self.m = self.m[1:] + [m_t]
        order = min(self.order, len(self.timesteps) - i, self.lower + 1)
        self.this_order = order
        self.last = x
        nxt = self._update(x, m_t, self.m[-2] if self.order > 1 else None, sig[i + 1], sig[i],
                           sig[i - 1] if i >= 1 else None, order)
        self.lower = min(self.lower + 1, self.order)
        self.i += 1
        return nxt


def sample_loop(velocity_fn, x, num_steps, shift, guide, solver="unipc"):
    """text2video.py:204-252 with velocity_fn(x, t) -> (cond, uncond)."""
    sch = UniPCOracle(num_steps, shift) if solver == "unipc" else DPMSolverOracle(num_steps, shift)
    for t in sch.timesteps:
        c, u = velocity_fn(x, t)
        x = sch.step(u + guide * (c - u), x)
    return x


# --------------------------------------------------------------------------------------------------
# Flow DPM-Solver++ (2M, midpoint) — seaweed_apt/wan/utils/fm_solvers.py, the sample_solver='dpm++'
# branch of WanT2V.generate (text2video.py:212-221).  Pinned against the reference class by
# oracle/make_golden.py -> tests/golden/dpmpp_6steps.npz.
# --------------------------------------------------------------------------------------------------
def dpm_sampling_sigmas(num_steps: int, shift: float, num_train_timesteps: int = 1000):
    """get_sampling_sigmas (fm_solvers.py:22-26) fed to set_timesteps(sigmas=...) (:226-290) of a scheduler
    constructed with shift=1: float32 sigmas [n+1] (first exactly 1, last 0), int64 (truncated) timesteps [n]."""
    s = np.linspace(1, 0, num_steps + 1)[:num_steps]
    s = shift * s / (1 + (shift - 1) * s)
    s = 1.0 * s / (1 + (1.0 - 1) * s)                       # the scheduler's own shift (=1)
    timesteps = torch.from_numpy(s * num_train_timesteps).to(torch.int64)
    sig = torch.from_numpy(np.concatenate([s, [0]]).astype(np.float32))
    return sig, timesteps


def dpm_default_sigmas(num_steps: int, shift: float, num_train_timesteps: int = 1000):
    """``set_timesteps(num_inference_steps)`` WITHOUT explicit sigmas (fm_solvers.py:226-290) of a scheduler constructed
    with ``shift`` (:177-190) — the call Omnihuman/omnihuman_wan_t2v.py:172-180,383 makes: the constructor's shifted
    table gives sigma_max / sigma_min, the linspace between them is shifted once more by config.shift."""
    alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
    base = torch.from_numpy(1.0 - alphas).to(torch.float32)
    base = shift * base / (1 + (shift - 1) * base)
    smax, smin = base[0].item(), base[-1].item()
    s = np.linspace(smax, smin, num_steps + 1).copy()[:-1]
    s = shift * s / (1 + (shift - 1) * s)
    timesteps = torch.from_numpy(s * num_train_timesteps).to(torch.int64)
    sig = torch.from_numpy(np.concatenate([s, [0]]).astype(np.float32))
    return sig, timesteps


class DPMSolverOracle:
    """solver_order 2, dpmsolver++, midpoint, flow_prediction, lower_order_final, final sigma 0.
    ``default_schedule``: the sigma table of ``set_timesteps(n)`` (dpm_default_sigmas) instead of the
    ``get_sampling_sigmas`` one WanT2V.generate feeds in."""

    def __init__(self, num_steps: int, shift: float, default_schedule: bool = False):
        self.sigmas, self.timesteps = (dpm_default_sigmas if default_schedule else dpm_sampling_sigmas)(num_steps, shift)
        self.m = [None, None]                               # x0 predictions, oldest first
        self.lower = 0
        self.i = 0

    @staticmethod
    def _lam(s):
        return torch.log(1 - s) - torch.log(s)              # log(alpha) - log(sigma)    (:333-334,459-460)

    def step(self, v: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """One scheduler.step(v, t_i, x) (:706-798)."""
        i, sig, n = self.i, self.sigmas, len(self.timesteps)
        lower_order_final = (i == n - 1)                    # final_sigmas_type == "zero"       (:745-748)
        lower_order_second = (i == n - 2) and n < 15        # lower_order_final and few steps   (:749-751)
        m0 = x - sig[i] * v                                 # convert_model_output              (:377-379)
        self.m = [self.m[1], m0]
        s_t, s_0 = sig[i + 1], sig[i]
        a_t = 1 - s_t
        h = self._lam(s_t) - self._lam(s_0)
        if self.lower < 1 or lower_order_final:             # first-order update                (:456-467)
            nxt = (s_t / s_0) * x - (a_t * (torch.exp(-h) - 1.0)) * m0
        else:                                               # 2M midpoint                       (:528-556)
            _ = lower_order_second                          # (order 2 either way)
            s_1 = sig[i - 1]
            h_0 = self._lam(s_0) - self._lam(s_1)
            r0 = h_0 / h
            D0, D1 = m0, (1.0 / r0) * (m0 - self.m[0])
            nxt = ((s_t / s_0) * x - (a_t * (torch.exp(-h) - 1.0)) * D0 - 0.5 * (a_t * (torch.exp(-h) - 1.0)) * D1)
        if self.lower < 2:
            self.lower += 1
        self.i += 1
        return nxt
