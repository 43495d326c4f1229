// Developer / test knobs of the library (kernel-selection thresholds, ordering choices of the sparse analysis, ...).
//
// A production library must not change its numerics or its kernel selection because of a stray environment variable, so the
// knobs are NOT read from the environment: dev_knob(name) returns the value set through mi355kkt_test_set_knob()
// (include/mi355kkt_test.h: an explicit call by a test or a developer script) or nullptr.  Only -DMI355KKT_DEBUG builds fall back
// to getenv(name).  The one environment variable every build honours is MI355KKT_ROCTX (profiler ranges; no effect on results).
#pragma once

namespace mi355kkt {

// nullptr: not set.  The returned string is a copy owned by the calling thread, valid until that thread's next dev_knob() call.
// Knobs are read where a plan / an engine is CREATED (or once per call): nothing is latched in function-local statics, so a
// later mi355kkt_test_set_knob() -- or the reset between two tests -- always takes effect.
const char* dev_knob(const char* name);
int set_dev_knob(const char* name, const char* value);     // value == nullptr: unset; name == nullptr: unset all

}  // namespace mi355kkt
