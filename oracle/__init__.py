"""CPU oracle for the AVID/CMA training step.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``avid-cma_amd/``) may
import from this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and there only as the checker / the
timed CPU baseline.
"""
