"""CPU oracle for the offline-MPC hot path -- TEST INFRASTRUCTURE ONLY.

Plain numpy/scipy fp64 restatements of the reference algorithm
(pratyushkumar211/industrial_nnmpc_2021, lib/linearMPC.py,
lib/controller_evaluation.py, lib/LinearMPCLayers.py).  Nothing under
``industrial_nnmpc_2021_amd/`` may import this package: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker / the CPU baseline, never as the thing shipped.

PARITY PINNING
--------------
* condensing / augmentation / chain driver / NN forward (SURVEY rows a1-a7,
  a9-a11): pinned against outputs of the reference itself, imported in the
  build container with ``cvxopt``/``h5py`` stubbed -- see
  ``tests/golden/make_golden.py`` and the ``.npz`` files it wrote.
* the QP arithmetic itself (row a8, ``cvxopt.solvers.qp``): the reference
  delegates it to cvxopt (un-vendored, un-pinned, absent here and no network)
  and holds no tests or golden vectors => **parity unpinned** at that seam.
  ``oracle.qp.coneqp_l`` restates cvxopt's published ``coneqp`` algorithm for
  the 'l' cone (used for the CPU baseline / stopping behaviour);
  correctness is anchored on the *exact optimum* (``oracle.qp.solve_exact``:
  tight fp64 PDIP + active-set polish, KKT-verified, cross-checked against
  scipy BVLS in the tests), which is unique because P > 0.
"""
