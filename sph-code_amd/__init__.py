"""sphx: MI355X (gfx950) implementation of the SPH inner loop of dmuley/sph-code.

  sph_code_amd.compat   drop-in for the reference's `navier_stokes_cleaned` hot-path callables
  sph_code_amd.sim      device-resident step loop (search -> dt -> sums -> leapfrog)
  sph_code_amd.ics      seeded initial conditions of the BASELINE configs
  sph_code_amd.build    hipcc driver for libsphx.so
There is no CPU fallback: every entry point raises if libsphx.so or a GPU is missing.
"""
__version__ = "0.1.0"
