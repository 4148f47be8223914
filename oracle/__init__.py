"""CPU oracle for the AlignModel hot path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package lyricalignment_amd never imports this.
"""
