// Tuning knobs of the kernel library.
//
// The PRODUCT library (librba_hip.so) has no writable global state besides the documented launch-geometry hint of rba_set_concurrent_streams
// (include/rba_hip.h): every knob below is a compile-time constant there (`static constexpr int`, no symbol, no branch that survives constant folding).
// The KNOBS build of the same sources (-DRBA_TUNE_KNOBS: librba_hip_knobs.so, `python -m rba_amd.csrc.build --knobs`) and the tools-only librba_tune.so
// turn them into exported ints that tools and tests may write to select a kernel variant (tests load that library through the `knobs` fixture of
// tests/conftest.py, tools through RBA_HIP_LIB / tools/_knobs.py).  Same entry points, same default dispatch, bit-identical results at the defaults.
//
//   RBA_KNOB(name, v)          one translation unit uses the knob: declaration + (knobs build) exported definition
//   RBA_KNOB_EXTERN(name, v)   in a header shared by several translation units
//   RBA_KNOB_DEFINE(name, v)   the one definition of a RBA_KNOB_EXTERN knob (nothing in the product build)
#pragma once
#ifdef RBA_TUNE_KNOBS
#define RBA_KNOB(name, v) extern "C" __attribute__((visibility("default"))) int name = v
#define RBA_KNOB_EXTERN(name, v) extern "C" int name
#define RBA_KNOB_DEFINE(name, v) extern "C" __attribute__((visibility("default"))) int name = v
#else
#define RBA_KNOB(name, v) static constexpr int name = v
#define RBA_KNOB_EXTERN(name, v) static constexpr int name = v
#define RBA_KNOB_DEFINE(name, v) static_assert(name == v, "product value of a knob")
#endif
