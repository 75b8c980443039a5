"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MoCo-Flow volume-rendering path.

Nothing under ``oracle/`` is part of the product. Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the reported CPU baseline. The shipped path
(``moco_flow_amd``) never imports this package and fails loudly when its HIP
library is missing.
"""
