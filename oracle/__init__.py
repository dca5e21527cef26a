"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the SUG hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  The product
(``sug_amd``) never imports this package and fails loudly without its HIP
library.
"""
