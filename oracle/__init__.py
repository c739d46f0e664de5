"""CPU restatement ("oracle") of the reference's RPO hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and there only as the checker / the timed CPU baseline, never as the thing shipped: the product
package ``rpo_amd`` does not import it and fails loudly when its HIP library is missing.

Parity status: **pinned** for CartSafe-v0 and SpringPendulum-v0 with RPODDPG / RPOSAC -- every function
here is checked against golden vectors produced by importing the unmodified reference in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
The reference ships no tests or golden vectors of its own.

Numeric types follow the reference: environment dynamics in float64 (the reference's ``env.step`` runs on
Python floats / numpy float64), everything that the reference does with torch in float32.
"""
