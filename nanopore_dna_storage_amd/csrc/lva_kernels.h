// lva_kernels.h -- launchers of the HIP kernels in lva_kernels.hip (stream passed as void*).
#pragma once
#include "lva_device.h"

namespace lva {

// whole step with the exact (reference-order) kernel
int launch_step_exact(const StepArgs& a, const Geometry& g, const DevCode* codes, uint32_t* trellis, void* stream);
// whole step with the fast kernel, followed by the exact fix-up pass over its work list
// ev_mid (a hipEvent_t or nullptr) is recorded between the fast kernel and the fix-up pass
int launch_step_fast(const StepArgs& a, const Geometry& g, const DevCode* codes, uint32_t* trellis, WorkHdr* hdr,
                     uint32_t* items, void* stream, void* ev_mid);
bool fast_kernel_available(const Geometry& g);
// whole step with the wavefront-per-target literal merge (2 <= L <= 64)
int launch_step_wave(const StepArgs& a, const Geometry& g, const DevCode* codes, uint32_t* trellis, void* stream);
bool wave_kernel_available(const Geometry& g);
// this launch's SlotStep records (a.steps), to be enqueued right before the step launch
int launch_prepare_step(const StepArgs& a, const DevCode* codes, SlotStep* steps, void* stream);
// initial scores of a slot + its descriptor (the read enters the slot)
int launch_init_slot(const Geometry& g, const DevCode* codes, uint32_t* trellis, uint32_t slot, const SlotDesc& desc,
                     SlotDesc* slots, void* stream);
int launch_gather_final(const Geometry& g, const DevCode* codes, const uint32_t* trellis, const GatherArgs& a,
                        uint32_t* results, void* stream);

}  // namespace lva
