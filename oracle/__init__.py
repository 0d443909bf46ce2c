"""CPU oracle for the detect -> align -> embed -> match hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product package
(``facerecognition_infrenceengine_amd``) never imports this package and has no
CPU fallback.

Pinning status (see DESIGN.md "Oracle"):

* match / decision / gallery-row / enrolment / unknown-cluster arithmetic
  (``oracle.match``, ``oracle.enrol``): PINNED.  ``tests/golden/make_golden.py``
  executes the reference's own methods (AST-extracted from
  ``/root/reference`` with the third-party model and the database mocked) and
  commits their inputs/outputs as ``tests/golden/*.npz``; ``tests/test_oracle_*``
  check this restatement against those vectors.
* detect / align / embed networks (``oracle.nets``, ``oracle.align``,
  ``oracle.detect``): PARITY UNPINNED.  The reference delegates them to the
  third-party ``insightface`` package (no version pin, model pack fetched from
  the network, absent here: SURVEY.md F2/F3), so there is no reference output to
  pin against.  They restate the published MTCNN / IResNet / 5-point-alignment
  algorithms in plain torch-CPU fp32 / numpy fp64 and are the *definition* the
  HIP kernels are judged against.
"""
