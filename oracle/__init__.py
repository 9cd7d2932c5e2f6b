"""CPU oracle for the vLGP variational-EM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the
product package ``vlgp_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg use it, and there only as the checker /
the timed CPU baseline.
"""
