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
int prof_begin(const char* name, double work);  // slot (>= 0) when this launch is timed, -1 otherwise
void prof_end(int slot);

struct ProfScope {
    int slot;
    ProfScope(const char* name, double work) : slot(prof_active() ? prof_begin(name, work) : -1) {}
    ~ProfScope() { if (slot >= 0) prof_end(slot); }
};
