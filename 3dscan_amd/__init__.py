"""MI355X-native decode -> unwrap -> correspond -> triangulate path of a Gray-code +
phase-shift structured-light scanner (drop-in for stages 3/4/5/7 of pranavkantgaur/3dscan).

The product is the C-ABI library built from csrc/ (see include/sl3d.h); this package is the
thin Python host layer around it (ctypes bindings, synthetic captures, multi-GPU launch glue).
The directory name starts with a digit, so import it with importlib.import_module("3dscan_amd").
"""
__version__ = "0.6.0"  # = SL3D_VERSION_STRING of include/sl3d.h (tests/test_abi.py checks)
