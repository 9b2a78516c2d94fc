"""CPU oracle for the neural-importance-sampling hot path (TEST INFRASTRUCTURE ONLY).

A plain NumPy (float64) restatement of the algorithm of NGoetz/NF (`nisrep`):
coupling-cell flow (PWQuad / PWLin + BatchNorm'd MLP conditioner + permutation
layers), moment estimators, and the "RAMBO on diet" phase-space map.  Every
function cites the reference file:line it follows.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here
against golden vectors produced by running the unmodified reference in the
build container (`tests/golden/gen_golden.py`, data in `tests/golden/*.npz`).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package.  The product (`nf_amd/`) never does: it
fails loudly when the HIP library or a GPU is missing.
"""
