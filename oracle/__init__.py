"""CPU oracle for the brute-force / IVF-Flat k-NN hot path.

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (vectordb-retrieval_amd/) never imports it.
"""
