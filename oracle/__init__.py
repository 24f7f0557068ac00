"""CPU oracle for the DODT per-frame hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (numpy) of the reference algorithms listed
in SURVEY.md section 8(a) and 8(f).
It exists to check the HIP path in ``dodt_amd``; it is never the thing that is
shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  ``dodt_amd`` must not.

Parity status (see DESIGN.md "Oracle pinning"):

* numpy half (a0-a6, a12, a14 numpy twins): PINNED.  Checked against golden
  vectors produced by importing the reference's own numpy code in the build
  container (``tests/golden/make_goldens.py``) and against the known answers
  in the reference's unit tests (``tests/test_oracle_kat.py``).
* TF-op half (a7-a11, a13: conv / batch-norm / pool / transposed conv /
  bilinear resize / crop_and_resize / non_max_suppression): PARITY UNPINNED.
  TensorFlow 1.3 is not installable here and no reference test asserts a
  numeric output of those ops; the restatement follows TF-1.3's documented
  semantics and is cross-checked against torch-CPU
  for the conv arithmetic only.
* detection records incl. the box_4ca heading correction (f3), KITTI rows, temporal module
  (f4): PINNED by fixtures the reference's own numpy code produced
  (``tests/golden/make_goldens_box4ca.py``, ``make_goldens_kitti.py``,
  ``make_goldens_temporal.py``).
"""
