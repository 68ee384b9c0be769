"""CPU oracle for the ferreus_bbfmm matvec hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / reported CPU
baseline.  The product (``ferreus_rbf_rs_amd``) never imports this package.

PARITY UNPINNED by the reference's own tests: the Rust reference
(`/root/reference`, ferreus_bbfmm + faer 0.23.2) can be neither compiled nor
imported in this environment (no cargo/rustc) and its tests hold no numeric
golden vector for the matvec (ferreus_bbfmm/src/bbfmm.rs:1464-1500 is an
error-path test; the doctests only print).  The oracle is therefore pinned
against (i) the dense O(N^2) direct sum it approximates and (ii) the
known-answer fixtures derivable from the reference text (see
``tests/golden/`` and ``tests/test_oracle_*.py``).
"""
