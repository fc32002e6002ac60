from .fm_solvers_unipc import FlowUniPCMultistepScheduler

__all__ = ["FlowUniPCMultistepScheduler"]
