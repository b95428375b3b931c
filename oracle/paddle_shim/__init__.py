"""A torch-backed stand-in for the subset of the ``paddle`` API that the reference's
synthesis-path modules use.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Purpose: PaddlePaddle cannot be installed in this environment, but the reference's own
Python source (parakeet/models/..., parakeet/modules/...) can still be *executed* if
``import paddle`` resolves to this package.  tools/make_golden.py does exactly that to
(1) validate the oracle's restatement against the reference's real code and (2) generate
the golden vectors under tests/golden/.  What this pins is Parakeet's logic (op order,
masks, transposes, buffer shuffles); Paddle's own kernel semantics are encoded here from
its documentation and stay "unverified" until a real Paddle run is available.
"""
