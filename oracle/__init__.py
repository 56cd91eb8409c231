"""TEST INFRASTRUCTURE ONLY - CPU oracle for the photonbend.core remap path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the reported CPU baseline.  The product
(`photonbend_amd`) never imports this package and fails loudly when its HIP
library is missing.

Parity status: PINNED.  ``oracle/make_goldens.py`` imports the real reference
(read-only, from /root/reference, in the build container only) and writes the
fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
restatement bit-for-bit against them.
"""
