"""reflector_ekf_slam_amd -- MI355X-native hot path of ShihanWang/reflector_ekf_slam.

Only what the path needs: ``csrc/`` (HIP kernels + the C ABI of include/*.h), the
host-side mirrors of the reference's two interfaces (``ekf_slam``, ``detect``), the
synthetic session generator (``synth``) and the session driver (``session``).
"""
from .ekf_slam import (DIFF, OMNI, EKFOptions, Map, Observation, OdometryData,  # noqa: F401
                       ReflectorEKFSLAM, ReflectorMatchResult, RekfError, State)
