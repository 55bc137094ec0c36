"""Parity oracle for the SMART hot path.  TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithm (ThibHlln/smartpy v0.2.2) used as the checker by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product package (smartpy_amd) must never
import anything from here.
"""
