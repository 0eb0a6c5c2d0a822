// internal: the 3x3 stride-1 weight-gradient kernel of conv_wgrad3.hip (called from conv_wgrad.hip, which owns the
// C-ABI entry points dlio_conv2d_wgrad / dlio_conv2d_wgrad_bf16 and the fixed-order slab reduction)
#pragma once
#include "common.h"

struct DlioWgrad3Plan {
  int ntb, mr, co_tiles, ci_chunks, tiles_w, tiles_h, splits;
  size_t ws_bytes;
};

// geometry only (also answers workspace queries); elem_bytes 4 = fp32 operands (split-bf16), 2 = bf16 operands
bool dlio_wgrad3_plan(const DlioConvDesc& d, int elem_bytes, DlioWgrad3Plan& p);
// writes p.splits slabs [Cout][Cin][3][3] to wsp; DLIO_EUNSUP when the pointers are not 16-byte aligned.
// amax_x / amax_dy (both or neither; elem_bytes 4): device floats holding the operands' largest magnitudes (or bounds on
// them) -- the products then run on the two-piece fp16 split
int dlio_wgrad3_launch(const void* x, const void* dy, float* wsp, const DlioConvDesc& d, const DlioWgrad3Plan& p,
                       int elem_bytes, hipStream_t s, const float* amax_x = nullptr, const float* amax_dy = nullptr);
