"""CPU oracle for the BSI hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Everything under ``oracle/`` is a CPU restatement (torch-CPU tensors, fp32 or
fp64) of the reference algorithm (martenlienen/bsi: ``bsi/bsi.py``,
``bsi/models/dit.py``, ``bsi/models/vdm_unet.py``, ``bsi/nn/*``).  It exists only
to check the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; nothing in ``bsi_amd/``
(the product) imports, calls or falls back to it.

Parity pinning: the restatement is checked (``tests/test_oracle_golden.py``)
against (i) the four known-answer tests of the reference
(``tests/test_bsi.py:7-34``, ``tests/models/components/test_fourier_features.py:9-28``)
and (ii) golden vectors generated in the build container by importing the
reference itself (``tools/gen_golden.py`` -> ``tests/golden/*.npz``; torch
2.10.0 CPU kernels).  The reference's own tests do not pin ``train_loss``,
``sample``, ``elbo`` or any model; for those the parity anchor is (ii).
"""
