"""CPU oracle for the Wan2.1 DiT / 3D-causal-VAE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker.  The
product path (``omnihuman-1-hack_amd``) never imports this package and fails
loudly when its HIP library is missing.

Contents
--------
detgen.py          deterministic, library-independent tensor generator
                   (splitmix64 counters) so weights/inputs can be regenerated
                   bit-identically on the GPU box instead of being shipped.
wan_dit_oracle.py  fp32 restatement of ``WanModel.forward``
                   (reference: seaweed_apt/wan/modules/model.py:502-588).
wan_vae_oracle.py  fp32 restatement of ``WanVAE_.encode/decode``
                   (reference: seaweed_apt/wan/modules/vae.py:516-568).
sampler_oracle.py  restatement of the flow-matching UniPC step
                   (reference: seaweed_apt/wan/utils/fm_solvers_unipc.py).
ref_import.py      shimmed import of the *real* reference modules; only works
                   where /root/reference exists (the build container).
make_golden.py     runs the real reference on detgen inputs and writes the
                   golden vectors committed under tests/golden/.

Pinning status: the reference holds no tests or golden vectors for this path
(SURVEY.md §4).  The oracle is therefore pinned against outputs of the
reference itself, imported in the build container by ``make_golden.py``; the
resulting vectors are committed under ``tests/golden/`` together with that
script.
"""
