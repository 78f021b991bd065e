"""Host-side mirror of the reference's src/tracker/core package: same names, argument meaning and
error behaviour; the arithmetic runs in the HIP kernels of csrc/kernels_trk.hip behind the C ABI."""
