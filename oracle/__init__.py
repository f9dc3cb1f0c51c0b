"""CPU oracle for the matching path -- test infrastructure only.

Importable from tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  The product package ``detr_amd`` never imports it.
"""
