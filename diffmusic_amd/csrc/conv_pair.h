// Fused convolution-pair launcher (conv_pair.hip) and the profiling hooks it shares with gemm_conv.hip.
#pragma once
#include <cstdlib>
#include <cstring>
#include "dmx_common.h"

bool dmx_conv_pair_eligible(const GemmDesc* a, const GemmDesc& b);
int dmx_conv_pair_launch(const GemmDesc* a, const GemmDesc& b, hipStream_t st);
// n (<= 3) mutually independent pairs of one width as a single grid, longest first (the branches of a HiFi-GAN resblock step)
int dmx_conv_pair_group_launch(int n, const GemmDesc* const* a, const GemmDesc* const* b, hipStream_t st);

// profiling records for launches that do not go through dmx_gemm_launch (no-ops unless dmx_prof_begin is active)
int dmx_prof_open(hipStream_t st);
void dmx_prof_close(int rec, hipStream_t st, double flops, double bytes, int M, int N, int K, int taps, int flags, int cfg);
