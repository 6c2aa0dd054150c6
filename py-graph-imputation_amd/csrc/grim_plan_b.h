// grim_plan_b.h -- Plan B / Plan C kernel (placeholder until the device path lands; subjects that
// need it are reported GRIM_ST_UNSUPPORTED, never silently computed elsewhere).
#pragma once
#include "grim_pair.h"
static int grim_launch_plan_b(DevArgs &A, uint32_t n_slots, hipStream_t stream) {
  (void)A; (void)n_slots; (void)stream;
  return 0;
}
