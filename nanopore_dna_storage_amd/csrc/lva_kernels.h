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
// initial scores of up to kTurnoverBatch slots + their descriptors (the reads enter their slots); no launch for an empty batch
int launch_init_slots(const Geometry& g, const DevCode* codes, uint32_t* trellis, const InitBatch& batch, SlotDesc* slots, void* stream);
// the final lists of up to kTurnoverBatch finished reads -> their result records
int launch_gather_finals(const Geometry& g, const DevCode* codes, const uint32_t* trellis, const GatherBatch& batch,
                         uint32_t* results, void* stream);

}  // namespace lva
