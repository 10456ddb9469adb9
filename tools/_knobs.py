"""Import FIRST in a tool that writes a tuning knob (csrc/knobs.h): makes rba_amd load the knobs build of the kernel library (librba_hip_knobs.so, built by
`python -m rba_amd.csrc.build --knobs`) unless RBA_HIP_LIB already names a library.  The product library has the knobs as compile-time constants."""
import os

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RBA_HIP_LIB", os.path.join(_REPO, "rba_amd", "csrc", "librba_hip_knobs.so"))
