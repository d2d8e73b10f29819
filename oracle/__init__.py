"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the SP/AT/LF hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / CPU baseline -- never as the thing shipped.  The
product package (``egocentric-gaze-prediction_amd``) must not import it.

The oracle is a plain torch-CPU / numpy restatement of the reference
algorithm; every function cites the reference ``file:line`` it follows.
Parity is pinned: ``tests/golden/*.npz`` were produced by importing the real
reference from ``/root/reference`` (script: ``tests/golden/make_golden.py``)
and ``tests/test_oracle_golden.py`` checks the oracle against them.
"""
