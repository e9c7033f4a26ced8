// LayerNorm-folding instantiations of the implicit-GEMM tiles (EPI_LNFOLD, gemm_tile.h): the U-Net transformer blocks' LN -> QKV / Q / FF1
// pairs (diffusers BasicTransformerBlock: norm1 -> attn1, norm2 -> attn2, norm3 -> ff; reached from the reference through
// diffmusic/pipelines/pipeline_musicldm.py:696-703).  A translation unit of its own so that it compiles next to gemm_conv.hip.
#include "gemm_tile.h"

namespace {
template <int BM, int BN, int WM, int WN, int NSTAGE>
int ln_glds(const GemmDesc& d, hipStream_t stream) {
  return (d.flags & EPI_GEGLU) ? launch_glds_t<BM, BN, WM, WN, NSTAGE, 3, true>(d, stream) : launch_glds_t<BM, BN, WM, WN, NSTAGE, 0, true>(d, stream);
}
template <int BM, int BN, int WM, int WN>
int ln_cfg(const GemmDesc& d, hipStream_t stream) {
  return (d.flags & EPI_GEGLU) ? launch_cfg_t<BM, BN, WM, WN, 3, true>(d, stream) : launch_cfg_t<BM, BN, WM, WN, 0, true>(d, stream);
}
// producers of LayerNorm inputs (EPI_ROWSTATS): the same tiles with the row-statistics epilogue
template <int BM, int BN, int WM, int WN, int NSTAGE>
int rs_glds(const GemmDesc& d, hipStream_t stream) { return launch_glds_t<BM, BN, WM, WN, NSTAGE, 4, false>(d, stream); }
template <int BM, int BN, int WM, int WN>
int rs_cfg(const GemmDesc& d, hipStream_t stream) { return launch_cfg_t<BM, BN, WM, WN, 4, false>(d, stream); }
}  // namespace

int dmx_gemm_launch_rowstats(int cfg, const GemmDesc& d, hipStream_t stream) {
  switch (cfg) {
    case 1: case 7: case 8: return rs_glds<256, 256, 2, 4, 2>(d, stream);
    case 2: case 9: case 19: return rs_glds<256, 128, 4, 2, 3>(d, stream);
    case 10: return rs_glds<192, 128, 4, 2, 3>(d, stream);
    case 11: return rs_glds<128, 128, 2, 2, 2>(d, stream);
    case 18: return rs_glds<128, 128, 2, 2, 4>(d, stream);
    case 12: case 15: return rs_glds<64, 64, 2, 2, 4>(d, stream);
    case 13: case 17: return rs_glds<128, 64, 2, 2, 3>(d, stream);
    case 14: case 16: return rs_glds<64, 128, 2, 2, 3>(d, stream);
    case 3: return rs_cfg<128, 128, 2, 2>(d, stream);
    case 4: case 5: return rs_cfg<128, 64, 2, 2>(d, stream);
    default: return rs_cfg<64, 64, 2, 2>(d, stream);
  }
}

int dmx_gemm_launch_ln(int cfg, const GemmDesc& d, hipStream_t stream) {
  // what the fold assumes: one tap, K = the normalised width, 16-bit output, statistics from the producer of A
  if (d.ntaps != 1 || d.K != d.Ci || d.ksplit > 1 || !d.colsum || !d.rowstats_in || d.nslots < 1 ||
      (d.flags & ~(EPI_LNFOLD | EPI_BIAS | EPI_GEGLU | EPI_RESID | EPI_ROWBIAS)))
    return DMX_ERR_SHAPE;
  switch (cfg) {
    case 1: case 7: case 8: return ln_glds<256, 256, 2, 4, 2>(d, stream);
    case 2: case 9: case 19: return ln_glds<256, 128, 4, 2, 3>(d, stream);
    case 10: return ln_glds<192, 128, 4, 2, 3>(d, stream);
    case 11: return ln_glds<128, 128, 2, 2, 2>(d, stream);
    case 18: return ln_glds<128, 128, 2, 2, 4>(d, stream);
    case 12: case 15: return ln_glds<64, 64, 2, 2, 4>(d, stream);
    case 13: case 17: return ln_glds<128, 64, 2, 2, 3>(d, stream);
    case 14: case 16: return ln_glds<64, 128, 2, 2, 3>(d, stream);
    case 3: return ln_cfg<128, 128, 2, 2>(d, stream);
    case 4: case 5: return ln_cfg<128, 64, 2, 2>(d, stream);
    default: return ln_cfg<64, 64, 2, 2>(d, stream);
  }
}
