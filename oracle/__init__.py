"""CPU oracle for the NeSVoR INR-training hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch-CPU / NumPy, fp32 or fp64) of the
algorithms on the path named by BASELINE.json:north_star.  It exists to *check*
the HIP kernels in ``nesvor_amd``; nothing in ``nesvor_amd`` imports it.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package.

Pinning status (see DESIGN.md "Oracle"):

* ``transform_convert`` — restates
  ``nesvor/transform/transform_convert_cuda_kernel.cu:14-440``; pinned against
  the reference's own 11-case scipy table (``tests/__init__.py:24-36``) and by
  fp64 ``gradcheck`` of the analytic backward formulas.
* ``slice_acq`` — restates ``nesvor/slice_acquisition/slice_acq_cuda_kernel.cu
  :17-171``; pinned by adjointness/known-answer tests and by the reference's
  Python wrappers run on top of it (golden fixtures).
* ``nesvor_model`` / ``train_loop`` — restate ``nesvor/nesvor/models.py`` and
  ``nesvor/nesvor/train.py``; pinned by golden vectors captured from the
  reference's Python (stub-imported in the build container, see
  ``tests/golden/make_golden.py``).
* ``hashgrid`` — **PARITY UNPINNED**.  The arithmetic lives in the external,
  un-vendored, un-pinned ``tinycudann`` (NVlabs/tiny-cuda-nn, call sites
  ``nesvor/nesvor/models.py:25,31``; ``requirements.txt:5`` names no commit).
  Its source is not under /root/reference; the oracle restates the published
  Instant-NGP / tiny-cuda-nn multi-resolution hash-grid algorithm and is
  self-checked (analytic grads == autograd of a naive gather formulation).
"""
