from .fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
from .fm_solvers_unipc import FlowUniPCMultistepScheduler

__all__ = ["FlowUniPCMultistepScheduler", "FlowDPMSolverMultistepScheduler", "get_sampling_sigmas",
           "retrieve_timesteps"]
