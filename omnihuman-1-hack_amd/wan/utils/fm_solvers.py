"""Flow-matching DPM-Solver++ (2M) scheduler — drop-in for the reference's
``FlowDPMSolverMultistepScheduler``, ``get_sampling_sigmas`` and
``retrieve_timesteps`` (seaweed_apt/wan/utils/fm_solvers.py), for the
configuration ``WanT2V.generate(sample_solver='dpm++')`` uses
(text2video.py:212-221): solver_order 2, dpmsolver++, midpoint,
flow_prediction, lower_order_final, final sigma 0, no thresholding.

A DPM-Solver++ step is ``x_next = c_x x + c_0 m0 + c_1 m1`` with
``m0 = x - sigma v`` the current and ``m1`` the previous x0 prediction; the
three scalars depend only on the sigma schedule.  They are computed on the host
in float32 with the reference's expressions (fm_solvers.py:456-467, 528-556;
the first sigma is exactly 1, so lambda = -inf there and the infinities cancel
the same way) and applied, together with classifier-free guidance and the x0
conversion, by the same single-pass kernel the UniPC scheduler uses
(``omh_cfg_unipc_step`` with the corrector off).
"""
import inspect
from typing import Optional, Tuple, Union

import numpy as np
import torch

from .._backend import ops

__all__ = ["FlowDPMSolverMultistepScheduler", "get_sampling_sigmas", "retrieve_timesteps", "dpmpp_coefficients"]


def get_sampling_sigmas(sampling_steps, shift):
    """fm_solvers.py:22-26."""
    sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
    return shift * sigma / (1 + (shift - 1) * sigma)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    """fm_solvers.py:29-66: forwards custom timesteps / sigmas to ``scheduler.set_timesteps``."""
    if timesteps is not None and sigmas is not None:
        raise ValueError("Only one of `timesteps` or `sigmas` can be passed. Please choose one to set custom values")
    params = set(inspect.signature(scheduler.set_timesteps).parameters.keys())
    if timesteps is not None:
        if "timesteps" not in params:
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support "
                             f"custom timestep schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(timesteps=timesteps, device=device, **kwargs)
    elif sigmas is not None:
        if "sigmas" not in params:
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support "
                             f"custom sigmas schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
        return scheduler.timesteps, num_inference_steps
    return scheduler.timesteps, len(scheduler.timesteps)


def _lam(sigma: torch.Tensor) -> torch.Tensor:
    return torch.log(1 - sigma) - torch.log(sigma)


def dpmpp_coefficients(sigmas: torch.Tensor, i: int, order: int) -> Tuple[float, Tuple[float, float, float]]:
    """(sigma_i, (c_x, c_0, c_1)) of step i: ``x_next = c_x x + c_0 m0 + c_1 m1``."""
    s_t, s_0 = sigmas[i + 1], sigmas[i]
    a_t = 1 - s_t
    h = _lam(s_t) - _lam(s_0)
    A = a_t * (torch.exp(-h) - 1.0)
    c_x = s_t / s_0
    if order == 1:                                          # fm_solvers.py:456-467
        return float(s_0), (float(c_x), float(-A), 0.0)
    h_0 = _lam(s_0) - _lam(sigmas[i - 1])                   # fm_solvers.py:528-556 (midpoint)
    inv_r0 = 1.0 / (h_0 / h)
    return float(s_0), (float(c_x), float(-A - 0.5 * A * inv_r0), float(0.5 * A * inv_r0))


class _SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class FlowDPMSolverMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2,
                 prediction_type: str = "flow_prediction", shift: Optional[float] = 1.0, use_dynamic_shifting=False,
                 thresholding: bool = False, dynamic_thresholding_ratio: float = 0.995, sample_max_value: float = 1.0,
                 algorithm_type: str = "dpmsolver++", solver_type: str = "midpoint", lower_order_final: bool = True,
                 euler_at_final: bool = False, final_sigmas_type: Optional[str] = "zero",
                 lambda_min_clipped: float = -float("inf"), variance_type: Optional[str] = None,
                 invert_sigmas: bool = False):
        if (solver_order not in (1, 2) or prediction_type != "flow_prediction" or use_dynamic_shifting
                or thresholding or algorithm_type != "dpmsolver++" or solver_type != "midpoint"
                or final_sigmas_type != "zero"):
            raise NotImplementedError("only the configuration WanT2V.generate uses is built (order<=2, dpmsolver++, "
                                      "midpoint, flow_prediction, final sigma 0)")

        class _Cfg(dict):
            __getattr__ = dict.__getitem__
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, solver_order=solver_order,
                           prediction_type=prediction_type, shift=shift, use_dynamic_shifting=False,
                           thresholding=False, algorithm_type="dpmsolver++", solver_type="midpoint",
                           lower_order_final=lower_order_final, euler_at_final=euler_at_final,
                           final_sigmas_type="zero")
        self.num_inference_steps = None
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()   # fm_solvers.py:177-187
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
        """fm_solvers.py:226-290."""
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        sigmas = np.asarray(sigmas)
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
        ts = self.timesteps.cpu()                                   # fm_solvers.py:679-703 on the host copy
        t = int(timestep) if not isinstance(timestep, torch.Tensor) else int(timestep.item())
        idx = (ts == t).nonzero()
        self._step_index = int(idx[1 if len(idx) > 1 else 0])

    def step_cfg(self, cond: torch.Tensor, uncond: torch.Tensor, guide_scale: float, sample: torch.Tensor,
                 timestep=None) -> torch.Tensor:
        """CFG combine + one DPM-Solver++ step in a single kernel; returns the next sample.
        Equivalent to ``step(uncond + g*(cond-uncond), t, sample)`` (text2video.py:243-252)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after "
                             "creating the scheduler")
        if self._step_index is None:
            if timestep is None:
                self._step_index = self._begin_index or 0
            else:
                self._init_step_index(timestep)
        i, n = self._step_index, len(self.timesteps)
        lower_order_final = (i == n - 1)                            # final_sigmas_type == "zero" (fm_solvers.py:745-748)
        first = self.config.solver_order == 1 or self.lower_order_nums < 1 or lower_order_final
        sigma_i, coef = dpmpp_coefficients(self.sigmas, i, 1 if first else 2)
        shape = sample.shape
        x = sample.contiguous().float()
        m1 = None if first else self.model_outputs[-1]
        mt = self._free.pop() if self._free else torch.empty_like(x)
        x_next = torch.empty_like(x)
        ops.cfg_unipc_step(cond.contiguous().float().view(shape), uncond.contiguous().float().view(shape), x,
                           None, m1, None, mt, None, x_next, guide_scale, sigma_i, False, (0.0, 0.0, 0.0, 0.0), coef)
        old = self.model_outputs[0]                                 # rotate the history (fm_solvers.py:754-756)
        if old is not None and not any(old is m for m in self.model_outputs[1:]):
            self._free.append(old)
        self.model_outputs = self.model_outputs[1:] + [mt]
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return x_next

    def step(self, model_output: torch.Tensor, timestep: Union[int, torch.Tensor], sample: torch.Tensor,
             generator=None, variance_noise: Optional[torch.Tensor] = None,
             return_dict: bool = True) -> Union[_SchedulerOutput, Tuple]:
        """Reference signature (fm_solvers.py:706-798)."""
        prev = self.step_cfg(model_output, model_output, 1.0, sample, timestep=timestep).view(sample.shape)
        prev = prev.to(model_output.dtype)
        if not return_dict:
            return (prev,)
        return _SchedulerOutput(prev_sample=prev)

    def add_noise(self, original_samples, noise, timesteps):
        """fm_solvers.py:815-854 (flow matching: (1-sigma) x + sigma noise)."""
        sig = self.sigmas.to(original_samples.device, original_samples.dtype)
        ts = self.timesteps.to(original_samples.device)
        idx = [int((ts == t).nonzero()[0]) for t in timesteps.reshape(-1)]
        sigma = sig[idx].flatten()
        while sigma.dim() < original_samples.dim():
            sigma = sigma.unsqueeze(-1)
        return (1 - sigma) * original_samples + sigma * noise

    def __len__(self):
        return self.config.num_train_timesteps
