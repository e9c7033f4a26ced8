// GroupNorm-statistics instantiations of the implicit-GEMM tiles (EPI_GNSTATS, gemm_epilogue.h): the convolutions that PRODUCE a GroupNorm
// input also write its partial sums, so the consumer's GroupNorm runs without a statistics pass over the tensor (the VAE decoder's and the
// U-Net's resnets: GroupNorm -> SiLU -> conv3x3 chains, diffusers ResnetBlock2D; reached from the reference through `vae.decode` in
// diffmusic/schedulers/scheduling_dps.py:195-197 and the U-Net call in diffmusic/pipelines/pipeline_musicldm.py:696-703).
// A translation unit of its own so that it compiles next to gemm_conv.hip.
#include "gemm_tile.h"

namespace {
thread_local int g_last_tm = 0;
// Only tiles whose epilogue walks 64-row chunks of 64-column wave tiles carry statistics (gemm_epilogue.h): a slot is one such chunk and its
// sums come out bit-identical from every member of the family, so what a GroupNorm sees does not depend on which member the cost model
// picked for this M (i.e. on the batch size).  Other tile choices are mapped onto the nearest member.
constexpr int kGnSlotRows = 64;
template <int BM, int BN, int WM, int WN, int NSTAGE>
int gn_glds(const GemmDesc& d, hipStream_t stream) {
  static_assert((BM / WM) % kGnSlotRows == 0 && BN / WN == 64, "not a member of the statistics family");
  const long long P = (long long)d.Hq * d.Wq;
  if (P < kGnSlotRows || d.M % P != 0) {              // images smaller than a slot / rows that are no whole images -- no statistics
    GemmDesc q = d; q.flags &= ~(EPI_GNSTATS | EPI_GNBWD);
    return launch_glds_t<BM, BN, WM, WN, NSTAGE, 0, false>(q, stream);
  }
  g_last_tm = kGnSlotRows;
  if (d.flags & EPI_GNBWD) return launch_glds_t<BM, BN, WM, WN, NSTAGE, 6, false>(d, stream);
  return launch_glds_t<BM, BN, WM, WN, NSTAGE, 5, false>(d, stream);
}
int gn_cfg128(const GemmDesc& d, hipStream_t stream) {          // register-staged family member: 128 x 128, wave tiles of 64 x 64
  const long long P = (long long)d.Hq * d.Wq;
  if (P < kGnSlotRows || d.M % P != 0) {
    GemmDesc q = d; q.flags &= ~(EPI_GNSTATS | EPI_GNBWD);
    return launch_cfg_t<128, 128, 2, 2, 0, false>(q, stream);
  }
  g_last_tm = kGnSlotRows;
  if (d.flags & EPI_GNBWD) return launch_cfg_t<128, 128, 2, 2, 6, false>(d, stream);
  return launch_cfg_t<128, 128, 2, 2, 5, false>(d, stream);
}
}  // namespace

int dmx_gemm_last_tile_rows() { return g_last_tm; }
void dmx_gemm_reset_last_tile_rows() { g_last_tm = 0; }

int dmx_gemm_launch_gnstats(int cfg, const GemmDesc& d, hipStream_t stream) {
  switch (cfg) {
    case 1: case 7: case 8: return gn_glds<256, 256, 2, 4, 2>(d, stream);       // (320 / 192-row tiles: chunks of 32 / 48 rows -> the 256-row tile)
    case 2: case 9: case 10: return gn_glds<256, 128, 4, 2, 3>(d, stream);
    case 19: return gn_glds<512, 128, 4, 2, 2>(d, stream);
    case 11: return gn_glds<128, 128, 2, 2, 2>(d, stream);
    case 12: case 13: case 14: case 15: case 16: case 17: case 18: return gn_glds<128, 128, 2, 2, 4>(d, stream);
    default: return gn_cfg128(d, stream);
  }
}
