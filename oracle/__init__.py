"""CPU oracle for the treegp GP hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the reported CPU baseline.  The
product path (``treegp_amd``) never imports this package and fails loudly
when the HIP library is missing.
"""
