// Optional per-kernel timing with HIP events on the launch stream (off by default).
// A scope brackets one kernel launch; every `stride`-th scope of a given name is timed.
// A profile belongs to the streams it was started on (one model's), so several models may be driven from several
// host threads while one of them is being profiled.
#pragma once
#include "common.h"
#include <string>

// Starts a profile over launches on stream s0.
void prof_start(hipStream_t s0, int stride);
// Makes launches on stream `extra` part of the profile that owns stream s (no-op when none is running).
void prof_attach(hipStream_t s, hipStream_t extra);
// Stops the profile that owns stream s, waits for the device, and returns a JSON array of
// {"name", "launches", "timed", "ms_timed", "work", "work_timed"} objects ("[]" when none was running).
std::string prof_stop(hipStream_t s);
// Whether launches on stream s are being profiled.
bool prof_active(hipStream_t s);
// slot (>= 0) when this launch is timed, -1 otherwise; the events go onto the stream the kernel is launched on
int prof_begin(const char* name, double work, hipStream_t s);
void prof_end(int slot, hipStream_t s);
// Whether the next launch recorded under `name` on stream s will be timed (its launch count falls on the sampling stride).
bool prof_next_timed(const char* name, hipStream_t s);
// Adds algorithmic work to a kernel's totals after the fact, for launches whose work is only known on the device
// when they are issued: `total` over all of them, `timed` over those that were timed.
void prof_add_work(const char* name, double total, double timed, hipStream_t s);

struct ProfScope {
    int slot;
    hipStream_t s;
    ProfScope(const char* name, double work, hipStream_t stream) : slot(prof_active(stream) ? prof_begin(name, work, stream) : -1), s(stream) {}
    ~ProfScope() { if (slot >= 0) prof_end(slot, s); }
};
