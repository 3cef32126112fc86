"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference hot path (model/univtg.py forward + dense criterion,
Hungarian matcher, span decode / NMS).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import anything from this package; the product
(``univtg_amd``) must never import it and has no CPU fallback.
"""
