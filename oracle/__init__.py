"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the CaTGrasp grasp-scoring hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import, link or execute it, and only as
the checker / timed CPU baseline.  The product (``catgrasp_amd``) never imports it.
"""
