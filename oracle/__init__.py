"""TEST INFRASTRUCTURE ONLY.

`oracle/` holds a CPU restatement of the reference algorithm for the DPOT hot path.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
anything from this package, and only as the *checker* / reported baseline - never as the
thing that is shipped or measured as the product.  Nothing under `dpot_amd/` imports it.
"""
