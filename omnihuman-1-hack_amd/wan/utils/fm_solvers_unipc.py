"""Flow-matching UniPC multistep scheduler — drop-in for the reference's
``FlowUniPCMultistepScheduler`` (seaweed_apt/wan/utils/fm_solvers_unipc.py),
for the configuration ``WanT2V.generate`` uses (text2video.py:205-211):
solver_order 2, bh2, predict_x0, flow_prediction, lower_order_final,
final sigma 0, no thresholding, no dynamic shifting.

UniPC's predictor and corrector are linear combinations of at most four
latent-sized tensors (the running sample, the last corrected sample and the
last two x0 predictions) whose scalar coefficients depend only on the sigma
schedule.  The host computes the coefficients (float32, the same expressions
as fm_solvers_unipc.py:404-470,548-619); one gfx950 kernel
(``omh_cfg_unipc_step``) then applies classifier-free guidance, the x0
conversion, the corrector and the predictor in a single pass over the latent —
the reference issues ~15 elementwise launches and a ``.nonzero().item()`` sync
per step (fm_solvers_unipc.py:628-640).
"""
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from .._backend import ops

__all__ = ["FlowUniPCMultistepScheduler", "unipc_coefficients"]


def _lam(sigma: torch.Tensor) -> torch.Tensor:
    return torch.log(1 - sigma) - torch.log(sigma)


def _bh_terms(h: torch.Tensor, order: int):
    """h_phi_1, B_h and the b vector of fm_solvers_unipc.py:423-442 (bh2, predict_x0)."""
    hh = -h
    h_phi_1 = torch.expm1(hh)
    h_phi_k = h_phi_1 / hh - 1
    B_h = torch.expm1(hh)
    b, factorial_i = [], 1
    for i in range(1, order + 1):
        b.append(h_phi_k * factorial_i / B_h)
        factorial_i *= i + 1
        h_phi_k = h_phi_k / hh - 1 / factorial_i
    return h_phi_1, B_h, b


def unipc_coefficients(sigmas: torch.Tensor, i: int, order_p: int, order_c: Optional[int]):
    """Scalar coefficients of step ``i`` (sigmas: float32 [n+1], last = 0).

    Returns (sigma_i, corr, pred):
      corr = None or (c_last, c_m1, c_m2, c_mt):  x_c = c_last*last + c_m1*m[i-1] + c_m2*m[i-2] + c_mt*m_t
      pred = (p_x, p_mt, p_m1):                   x_next = p_x*x_c + p_mt*m_t + p_m1*m[i-1]
    """
    sig = sigmas.to(torch.float32)
    corr = None
    if order_c is not None:
        # corrector from s0 = sigma[i-1] to t = sigma[i]          (:548-619)
        s_t, s_0 = sig[i], sig[i - 1]
        a_t = 1 - s_t
        h = _lam(s_t) - _lam(s_0)
        h_phi_1, B_h, b = _bh_terms(h, order_c)
        if order_c == 1:
            rho_last, rho0_over_rk = torch.tensor(0.5), torch.tensor(0.0)
        else:
            rk = (_lam(sig[i - 2]) - _lam(s_0)) / h
            R = torch.stack([torch.stack([torch.tensor(1.0), torch.tensor(1.0)]),
                             torch.stack([rk, torch.tensor(1.0)])])
            rhos = torch.linalg.solve(R, torch.stack(b))
            rho_last, rho0_over_rk = rhos[1], rhos[0] / rk
        corr = (float(s_t / s_0),
                float(-a_t * h_phi_1 + a_t * B_h * (rho0_over_rk + rho_last)),
                float(-a_t * B_h * rho0_over_rk),
                float(-a_t * B_h * rho_last))
    # predictor from s0 = sigma[i] to t = sigma[i+1]               (:404-470)
    s_t, s_0 = sig[i + 1], sig[i]
    a_t = 1 - s_t
    h = _lam(s_t) - _lam(s_0)
    h_phi_1, B_h, _ = _bh_terms(h, order_p)
    if order_p == 1:
        pred = (float(s_t / s_0), float(-a_t * h_phi_1), 0.0)
    else:
        rk = (_lam(sig[i - 1]) - _lam(s_0)) / h
        half = 0.5 * a_t * B_h / rk
        pred = (float(s_t / s_0), float(-a_t * h_phi_1 + half), float(-half))
    return float(sig[i]), corr, pred


class _SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class FlowUniPCMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2,
                 prediction_type: str = "flow_prediction", shift: Optional[float] = 1.0, use_dynamic_shifting=False,
                 thresholding: bool = False, dynamic_thresholding_ratio: float = 0.995, sample_max_value: float = 1.0,
                 predict_x0: bool = True, solver_type: str = "bh2", lower_order_final: bool = True,
                 disable_corrector: List[int] = [], solver_p=None, timestep_spacing: str = "linspace",
                 steps_offset: int = 0, final_sigmas_type: Optional[str] = "zero"):
        if (solver_order not in (1, 2) or prediction_type != "flow_prediction" or use_dynamic_shifting
                or thresholding or not predict_x0 or solver_type != "bh2" or solver_p is not None
                or final_sigmas_type != "zero"):
            raise NotImplementedError("only the configuration WanT2V.generate uses is built "
                                      "(order<=2, bh2, predict_x0, flow_prediction, final sigma 0)")

        class _Cfg(dict):
            __getattr__ = dict.__getitem__
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, solver_order=solver_order,
                           prediction_type=prediction_type, shift=shift, use_dynamic_shifting=False,
                           thresholding=False, predict_x0=True, solver_type="bh2",
                           lower_order_final=lower_order_final, final_sigmas_type="zero")
        self.predict_x0 = True
        self.disable_corrector = list(disable_corrector)
        self.num_inference_steps = None
        # fm_solvers_unipc.py:106-131
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.sigmas = sigmas
        self.timesteps = sigmas * num_train_timesteps
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()
        self._reset()

    def _reset(self):
        self.model_outputs = [None] * self.config.solver_order     # x0 predictions, oldest first
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = None
        self._step_index = None
        self._begin_index = None
        self._free = []

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, shift=None):
        """fm_solvers_unipc.py:160-227."""
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        if shift is None:
            shift = self.config.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        timesteps = sigmas * self.config.num_train_timesteps
        sigmas = np.concatenate([sigmas, [0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self._reset()

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def _init_step_index(self, timestep):
        if self._begin_index is not None:
            self._step_index = self._begin_index
            return
        # fm_solvers_unipc.py:628-640, on the host copy of the (tiny) timestep table
        ts = self.timesteps.cpu()
        t = int(timestep) if not isinstance(timestep, torch.Tensor) else int(timestep.item())
        idx = (ts == t).nonzero()
        self._step_index = int(idx[1 if len(idx) > 1 else 0])

    # ------------------------------------------------------------------ stepping
    def step_cfg(self, cond: torch.Tensor, uncond: torch.Tensor, guide_scale: float, sample: torch.Tensor,
                 timestep=None) -> torch.Tensor:
        """CFG combine + one UniPC step in a single kernel; returns the next sample.
        Equivalent to ``step(uncond + g*(cond-uncond), t, sample)`` (text2video.py:243-252)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after "
                             "creating the scheduler")
        if self._step_index is None:
            if timestep is None:
                self._step_index = self._begin_index or 0
            else:
                self._init_step_index(timestep)
        i = self._step_index
        shape = sample.shape
        x = sample.contiguous().float()
        use_corr = (i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None)
        order_c = self.this_order if use_corr else None
        if self.config.lower_order_final:
            this_order = min(self.config.solver_order, len(self.timesteps) - i)
        else:
            this_order = self.config.solver_order
        this_order = min(this_order, self.lower_order_nums + 1)
        sigma_i, corr, pred = unipc_coefficients(self.sigmas, i, this_order, order_c)
        m1 = self.model_outputs[-1]
        m2 = self.model_outputs[-2] if self.config.solver_order > 1 else None
        mt = self._free.pop() if self._free else torch.empty_like(x)
        xc = self.last_sample if self.last_sample is not None else torch.empty_like(x)
        x_next = torch.empty_like(x)
        ops.cfg_unipc_step(cond.contiguous().float().view(shape), uncond.contiguous().float().view(shape), x,
                           self.last_sample, m1, m2, mt, xc, x_next, guide_scale, sigma_i, use_corr,
                           corr or (0.0, 0.0, 0.0, 0.0), pred)
        # rotate the x0-prediction history (fm_solvers_unipc.py:700-704)
        old = self.model_outputs[0]
        if old is not None and not any(old is m for m in self.model_outputs[1:]):
            self._free.append(old)
        self.model_outputs = self.model_outputs[1:] + [mt]
        self.this_order = this_order
        self.last_sample = xc
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return x_next

    def step(self, model_output: torch.Tensor, timestep: Union[int, torch.Tensor], sample: torch.Tensor,
             return_dict: bool = True, generator=None) -> Union[_SchedulerOutput, Tuple]:
        """Reference signature (fm_solvers_unipc.py:655-739)."""
        prev = self.step_cfg(model_output, model_output, 1.0, sample, timestep=timestep).view(sample.shape)
        prev = prev.to(sample.dtype)
        if not return_dict:
            return (prev,)
        return _SchedulerOutput(prev_sample=prev)

    def add_noise(self, original_samples, noise, timesteps):
        """fm_solvers_unipc.py:760-800 (flow matching: (1-sigma) x + sigma noise)."""
        sig = self.sigmas.to(original_samples.device, original_samples.dtype)
        ts = self.timesteps.to(original_samples.device)
        idx = [int((ts == t).nonzero()[0]) for t in timesteps.reshape(-1)]
        sigma = sig[idx].flatten()
        while sigma.dim() < original_samples.dim():
            sigma = sigma.unsqueeze(-1)
        return (1 - sigma) * original_samples + sigma * noise

    def __len__(self):
        return self.config.num_train_timesteps
