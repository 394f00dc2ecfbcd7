"""Synthetic rectified stereo pairs for the stereo-association tests."""
from orb_slam3_amd.synth import right_view  # noqa: F401
