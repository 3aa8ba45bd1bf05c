"""Shim: the un-vendored ``opensimplex`` dependency -> the repo's restatement.
Uses the C build (oracle/_build/libosimplex.so) when present, else pure Python; both are
bit-identical (tests/test_noise.py)."""
from oracle.noise import OpenSimplex  # noqa: F401
