// GroupNorm-statistics instantiations of the implicit-GEMM tiles (EPI_GNSTATS, gemm_epilogue.h): the convolutions that PRODUCE a GroupNorm
// input also write its partial sums, so the consumer's GroupNorm runs without a statistics pass over the tensor (the VAE decoder's and the
// U-Net's resnets: GroupNorm -> SiLU -> conv3x3 chains, diffusers ResnetBlock2D; reached from the reference through `vae.decode` in
// diffmusic/schedulers/scheduling_dps.py:195-197 and the U-Net call in diffmusic/pipelines/pipeline_musicldm.py:696-703).
// A translation unit of its own so that it compiles next to gemm_conv.hip.
#include "gemm_tile.h"

namespace {
thread_local int g_last_tm = 0;
template <int BM, int BN, int WM, int WN, int NSTAGE>
int gn_glds(const GemmDesc& d, hipStream_t stream) {
  if ((long long)d.Hq * d.Wq < BM / WM) {            // images smaller than a wave tile: slots could not tell the images apart -- no statistics
    GemmDesc q = d; q.flags &= ~(EPI_GNSTATS | EPI_GNBWD);
    return launch_glds_t<BM, BN, WM, WN, NSTAGE, 0, false>(q, stream);
  }
  g_last_tm = BM / WM;
  if (d.flags & EPI_GNBWD) return launch_glds_t<BM, BN, WM, WN, NSTAGE, 6, false>(d, stream);
  return launch_glds_t<BM, BN, WM, WN, NSTAGE, 5, false>(d, stream);
}
template <int BM, int BN, int WM, int WN>
int gn_cfg(const GemmDesc& d, hipStream_t stream) {
  if ((long long)d.Hq * d.Wq < BM / WM) {
    GemmDesc q = d; q.flags &= ~(EPI_GNSTATS | EPI_GNBWD);
    return launch_cfg_t<BM, BN, WM, WN, 0, false>(q, stream);
  }
  g_last_tm = BM / WM;
  if (d.flags & EPI_GNBWD) return launch_cfg_t<BM, BN, WM, WN, 6, false>(d, stream);
  return launch_cfg_t<BM, BN, WM, WN, 5, false>(d, stream);
}
}  // namespace

int dmx_gemm_last_tile_rows() { return g_last_tm; }
void dmx_gemm_reset_last_tile_rows() { g_last_tm = 0; }

int dmx_gemm_launch_gnstats(int cfg, const GemmDesc& d, hipStream_t stream) {
  switch (cfg) {
    case 1: return gn_glds<256, 256, 2, 4, 2>(d, stream);
    case 2: return gn_glds<256, 128, 4, 2, 3>(d, stream);
    case 7: return (d.flags & EPI_GNBWD) ? gn_glds<256, 256, 2, 4, 2>(d, stream)      // (the backward sums do not fit the 320-row tile's registers)
                                         : gn_glds<320, 256, 2, 4, 2>(d, stream);
    case 8: return gn_glds<192, 256, 2, 4, 2>(d, stream);
    case 9: return gn_glds<320, 128, 4, 2, 2>(d, stream);
    case 10: return gn_glds<192, 128, 4, 2, 3>(d, stream);
    case 11: return gn_glds<128, 128, 2, 2, 2>(d, stream);
    case 18: return gn_glds<128, 128, 2, 2, 4>(d, stream);
    case 12: case 15: return gn_glds<64, 64, 2, 2, 4>(d, stream);
    case 13: case 17: return gn_glds<128, 64, 2, 2, 3>(d, stream);
    case 14: case 16: return gn_glds<64, 128, 2, 2, 3>(d, stream);
    case 19: return gn_glds<512, 128, 4, 2, 2>(d, stream);
    case 3: return gn_cfg<128, 128, 2, 2>(d, stream);
    case 4: return gn_cfg<128, 64, 2, 2>(d, stream);
    case 5: return gn_cfg<128, 32, 4, 1>(d, stream);
    default: return gn_cfg<64, 64, 2, 2>(d, stream);
  }
}
