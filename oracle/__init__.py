"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the SAT train-step hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path lives in
``show-attend-and-tell-pytorch-lightning_amd/`` and fails loudly when its HIP
library is missing; it never routes through this package.
"""
