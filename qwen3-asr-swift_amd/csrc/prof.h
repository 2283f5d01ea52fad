// Optional per-kernel timing with HIP events on the launch stream (off by default).
// A scope brackets one kernel launch; every `stride`-th scope of a given name is timed.
#pragma once
#include "common.h"
#include <string>

void prof_start(hipStream_t s, int stride);
// Stops, waits for the stream, and returns a JSON array of
// {"name", "launches", "timed", "ms_timed", "work", "work_timed"} objects.
std::string prof_stop();
bool prof_active();
// slot (>= 0) when this launch is timed, -1 otherwise; the events go onto the stream the kernel is launched on
int prof_begin(const char* name, double work, hipStream_t s);
void prof_end(int slot, hipStream_t s);

struct ProfScope {
    int slot;
    hipStream_t s;
    ProfScope(const char* name, double work, hipStream_t stream) : slot(prof_active() ? prof_begin(name, work, stream) : -1), s(stream) {}
    ~ProfScope() { if (slot >= 0) prof_end(slot, s); }
};
